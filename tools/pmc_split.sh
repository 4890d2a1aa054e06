#!/bin/bash
# SQ counters of the split conv kernel (one shape), separate pass from any tracing
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ONLY=${ONLY:-6} REPS=3 MOTIF_CONV_MMA=6
run() {
    tag=$1; shift
    ( for kv in "$@"; do export "$kv"; done
      rocprofv3 --pmc $PMC -d $R/gpurun_out/pmc_$tag -o t --output-format csv -- python $R/tools/conv_bench.py > /dev/null 2>&1
      f=$(find $R/gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
      echo "== $tag"; python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "conv_split_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("  %-28s %14.0f  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
    )
}
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"
run full
run mfma_only MOTIF_CONV_DBG=13
PMC="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM"
run full2
