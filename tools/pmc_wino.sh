#!/bin/bash
# SQ counters of the round-4 Winograd conv kernel on one shape (counters only; two passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ONLY=${ONLY:-0} REPS=3 ENGINE=${ENGINE:-5}
KN=${KN:-conv_wino_kernel}
run() {
    tag=$1; PMC=$2
    rm -rf $R/gpurun_out/pmc_$tag
    rocprofv3 --pmc $PMC -d $R/gpurun_out/pmc_$tag -o t --output-format csv -- python3 $R/tools/conv_bench.py > /dev/null 2>&1
    f=$(find $R/gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
    echo "== $tag ($KN, shape $ONLY)"; python3 - "$f" "$KN" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("  %-28s %14.0f  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
    rm -rf $R/gpurun_out/pmc_$tag
}
run wsq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
run wsq2 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
