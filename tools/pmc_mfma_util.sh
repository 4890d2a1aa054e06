#!/bin/bash
# MFMA-pipe utilisation of every kernel of the c2 clip: one counters-only rocprofv3 pass over a short bench run.
# util = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 shader engines * 1024 SIMDs), clock = SQ_BUSY_CYCLES / 32 / duration is not
# available in a counters-only pass (no timestamps), so the table lists busy cycles per launch instead.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_util
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA \
    -d $R/gpurun_out/pmc_util -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --streams 1 --batch 1 --no-fp32-leg --no-cpu-baseline --no-roofline --no-pwc --no-streams1 > /dev/null 2>&1
python3 - $R/gpurun_out/pmc_util <<'PY'
import csv, sys, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CYCLES":
            cnt[k] += 1
print("%-46s %7s %12s %9s %8s %8s %8s %8s" % ("kernel", "calls", "kcyc/launch", "MFMA util", "wait", "iwait", "valu", "lds"))
rows = sorted(acc.items(), key=lambda kv: -kv[1]["SQ_BUSY_CYCLES"])
for k, m in rows[:16]:
    busy = m["SQ_BUSY_CYCLES"] / 32.0
    wc = m["SQ_WAVE_CYCLES"] or 1.0
    print("%-46s %7d %12.1f %9.3f %8.3f %8.3f %8.3f %8.3f" % (k, cnt[k], busy / max(cnt[k], 1) / 1e3, m["SQ_VALU_MFMA_BUSY_CYCLES"] / (busy * 1024) if busy else 0,
          m["SQ_WAIT_ANY"] / wc, m["SQ_WAIT_INST_ANY"] / wc, m["SQ_ACTIVE_INST_VALU"] / wc, m["SQ_ACTIVE_INST_LDS"] / wc))
PY
