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
# NOTE: TA_* / TCP_* counter passes hung rocprofv3 on this pool (900 s timeout, no output) -- SQ counters only.
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
run sq
