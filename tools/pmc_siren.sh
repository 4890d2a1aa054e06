#!/bin/bash
# SQ counters of the SIREN kernels (tools/siren_bench.py ONLY=split)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ONLY=split REPS=1
run() {
    tag=$1
    rocprofv3 --pmc $PMC -d $R/gpurun_out/pmcs_$tag -o t --output-format csv -- python $R/tools/siren_bench.py > /dev/null 2>&1
    f=$(find $R/gpurun_out/pmcs_$tag -name "*counter_collection.csv" | head -1)
    echo "== $tag"; python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "siren" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(" ", k)
    for c, v in sorted(d.items()):
        print("     %-28s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
}
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA"
run a
PMC="SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU"
run b
