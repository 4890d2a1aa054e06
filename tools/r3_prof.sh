#!/bin/bash
# kernel-level profile of the bench clip (single stream, no secondary legs): rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3
rm -rf $R/gpurun_out/r3/prof
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3/prof -o t -- python3 $R/bench.py --steps 8 --warmup 2 --streams 1 --no-fp32-leg --no-cpu-baseline > $R/gpurun_out/r3/prof_bench.json 2> $R/gpurun_out/r3/prof_bench.err
db=$(find $R/gpurun_out/r3/prof -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $db > $R/gpurun_out/r3/kernel_stats.txt 2>&1
head -70 $R/gpurun_out/r3/kernel_stats.txt
find $R/gpurun_out/r3/prof -name "*.db" -delete
