#!/bin/bash
# Round-3 counter passes (counters only, no tracing flags; each --pmc set in its own run):
#   1. MFMA-pipe utilisation + wave-time split of every kernel of the c2 clip  -> profiles/r03_mfma_util_by_kernel.txt
#   2. HBM-side traffic (FETCH_SIZE, WRITE_SIZE: separate passes) of every kernel -> profiles/r03_hbm_traffic_by_kernel.txt
#   3. the same two counters for the two 3x3 kernels on the recon-trunk launch    -> profiles/r03_conv_traffic.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
bash $R/tools/pmc_mfma_util_r02.sh > $O/r03_mfma_util_by_kernel.txt 2>&1
bash $R/tools/pmc_hbm_by_kernel_r02.sh > $O/r03_hbm_traffic_by_kernel.txt 2>&1
export ONLY=0 REPS=3
: > $O/r03_conv_traffic.txt
for e in 1 2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_conv_${e}_$c
    ENGINE=$e rocprofv3 --pmc $c -d $O/pmc_conv_${e}_$c -o t --output-format csv -- python3 $R/tools/conv_bench.py > /dev/null 2>&1
    f=$(find $O/pmc_conv_${e}_$c -name "*counter_collection.csv" | head -1)
    python3 - "$f" $c $e >> $O/r03_conv_traffic.txt <<'PY'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "conv_split" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]]
print("engine", sys.argv[3], sys.argv[2], "KB per launch: mean %.1f  n=%d" % (sum(v) / max(len(v), 1), len(v)))
PY
    rm -rf $O/pmc_conv_${e}_$c
  done
done
cat $O/r03_mfma_util_by_kernel.txt | tail -18; cat $O/r03_hbm_traffic_by_kernel.txt | tail -16; cat $O/r03_conv_traffic.txt
rm -rf $R/gpurun_out/pmc_util $R/gpurun_out/pmc_hbm_FETCH_SIZE $R/gpurun_out/pmc_hbm_WRITE_SIZE
