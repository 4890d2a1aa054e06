#!/bin/bash
# Kernel-level profile of the bench command (rocprofv3 --kernel-trace --stats, single stream, no secondary legs) -> gpurun_out/prof/
#   gpurun -- 'bash tools/prof_bench.sh'      then copy gpurun_out/prof/kernel_stats.txt to profiles/rNN_bench_kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
rm -rf $O/trace
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 $R/bench.py --steps 8 --warmup 2 --streams 1 --batch 1 --no-fp32-leg --no-cpu-baseline --no-pwc --no-streams1 > $O/prof_bench.json 2> $O/prof_bench.err
db=$(find $O/trace -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $db > $O/kernel_stats.txt 2>&1
head -45 $O/kernel_stats.txt
find $O/trace -name "*.db" -delete
