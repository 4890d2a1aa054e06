#!/bin/bash
# HBM traffic of the dominant conv kernel on the recon_trunk shape: FETCH_SIZE and WRITE_SIZE in separate --pmc passes
# (TCC slots: FETCH_SIZE costs 3, WRITE_SIZE 2 -- they do not fit one pass; no tracing flags beside --pmc).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ONLY=0 REPS=3
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $R/gpurun_out/pmct2_$c -o t --output-format csv -- python3 $R/tools/conv_bench.py > /dev/null 2>&1
  f=$(find $R/gpurun_out/pmct2_$c -name "*counter_collection.csv" | head -1)
  cp $f $R/gpurun_out/r02_conv_split_${c}_pmc.csv
  python3 - "$f" $c <<'PY'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "conv_split_kernel" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]]
print(sys.argv[2], "per launch: mean %.1f  n=%d" % (sum(v) / len(v), len(v)))
PY
done
