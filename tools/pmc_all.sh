#!/bin/bash
# Counter passes of a round (counters only, no tracing flags; each --pmc set in its own run):
#   1. MFMA-pipe utilisation + wave-time split of every kernel of the c2 clip  -> gpurun_out/prof/mfma_util_by_kernel.txt
#   2. HBM-side traffic (FETCH_SIZE, WRITE_SIZE: separate passes) of every kernel -> gpurun_out/prof/hbm_traffic_by_kernel.txt
#   3. the same two counters for the 3x3 kernels on the recon-trunk launch (conv_bench shape 0) -> gpurun_out/prof/conv_traffic.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
bash $R/tools/pmc_mfma_util.sh > $O/mfma_util_by_kernel.txt 2>&1
bash $R/tools/pmc_hbm_by_kernel.sh > $O/hbm_traffic_by_kernel.txt 2>&1
export ONLY=0 REPS=3
: > $O/conv_traffic.txt
for e in 5 2 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_conv_${e}_$c
    ENGINE=$e rocprofv3 --pmc $c -d $O/pmc_conv_${e}_$c -o t --output-format csv -- python3 $R/tools/conv_bench.py > /dev/null 2>&1
    f=$(find $O/pmc_conv_${e}_$c -name "*counter_collection.csv" | head -1)
    python3 - "$f" $c $e >> $O/conv_traffic.txt <<'PY'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if ("conv_split" in r["Kernel_Name"] or "conv_wino" in r["Kernel_Name"]) and "pack" not in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]]
print("engine", sys.argv[3], sys.argv[2], "KB per launch: mean %.1f  n=%d" % (sum(v) / max(len(v), 1), len(v)))
PY
    rm -rf $O/pmc_conv_${e}_$c
  done
done
tail -18 $O/mfma_util_by_kernel.txt; tail -16 $O/hbm_traffic_by_kernel.txt; cat $O/conv_traffic.txt
rm -rf $R/gpurun_out/pmc_util $R/gpurun_out/pmc_hbm_FETCH_SIZE $R/gpurun_out/pmc_hbm_WRITE_SIZE
