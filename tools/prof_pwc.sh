#!/bin/bash
# Kernel-level profile of PWC-Net (forward on 720x1280 pairs, tools/pwc_bench.py) -> gpurun_out/prof/pwc_kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
rm -rf $O/trace_pwc
rocprofv3 --kernel-trace --stats -d $O/trace_pwc -o t -- python3 $R/tools/pwc_bench.py > $O/pwc_bench.txt 2> $O/pwc_bench.err
db=$(find $O/trace_pwc -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $db > $O/pwc_kernel_stats.txt 2>&1
tail -8 $O/pwc_bench.txt; head -30 $O/pwc_kernel_stats.txt
find $O/trace_pwc -name "*.db" -delete
