#!/bin/bash
# kernel-level timing of the conv engines on one shape (ONLY=<shape index>) under rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ONLY=${ONLY:-0} REPS=30
run() {  # tag, env assignments...
    tag=$1; shift
    ( for kv in "$@"; do export "$kv"; done
      rocprofv3 --kernel-trace -d $R/gpurun_out/ps_$tag -o t -- python $R/tools/conv_bench.py > /dev/null 2>&1
      db=$(find $R/gpurun_out/ps_$tag -name "*.db" | head -1)
      echo "== $tag"; python $R/tools/rocpd_stats.py $db | grep -E "conv_(split|igemm)_kernel" | cut -c1-60,90-160 )
}
run x6 MOTIF_CONV_MMA=6
