#!/bin/bash
# SQ counters of conv_split_kernel<3,4> on one conv_bench shape (ONLY=<index>; 0 = the recon-trunk launch), two passes
# of <= 8 SQ counters each, counters only (no tracing flags).  Prints per-launch averages and derived ratios.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ONLY=${ONLY:-0} REPS=3 MOTIF_CONV_MMA=6
pass() {
    tag=$1; shift
    rm -rf $R/gpurun_out/pmc_$tag
    timeout 240 rocprofv3 --pmc "$@" -d $R/gpurun_out/pmc_$tag -o t --output-format csv -- python3 $R/tools/conv_bench.py > /dev/null 2>&1
}
pass sqa SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
pass sqb SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_WAVES
python3 - $R/gpurun_out/pmc_sqa $R/gpurun_out/pmc_sqb <<'PY'
import csv, sys, glob, collections
acc = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_split_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(m):
    print("  %-32s %16.0f  (n=%d)" % (k, m[k], len(acc[k])))
g = m.get
if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_BUSY_CYCLES"):
    print("  MFMA busy / (SQ busy cycles x 4 SIMDs... per-SE aggregation: see DESIGN) = %.3f" % (g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES")))
if g("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS"):
        if g(k): print("  %-24s / SQ_WAVE_CYCLES = %.3f" % (k, g(k) / g("SQ_WAVE_CYCLES")))
PY
