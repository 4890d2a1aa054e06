#!/bin/bash
# HBM-side traffic (FETCH_SIZE, WRITE_SIZE; two counters-only passes, no tracing flags) of every kernel of the c2 clip over a short bench run.
# FETCH_SIZE on gfx950 reports half the bytes of 16-byte-per-lane streams (MI355X_MICROARCH.md "HBM"); other widths are uncalibrated, so the
# table gives the raw counter and the doubled value side by side.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_hbm_$c
  timeout 400 rocprofv3 --pmc $c -d $R/gpurun_out/pmc_hbm_$c -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --streams 1 --batch 1 --no-fp32-leg --no-cpu-baseline --no-roofline --no-pwc --no-streams1 > /dev/null 2>&1
done
python3 - $R/gpurun_out/pmc_hbm_FETCH_SIZE $R/gpurun_out/pmc_hbm_WRITE_SIZE <<'PY'
import csv, sys, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
import json, os
doc = {"unit": "bytes per launch", "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KB), two counters-only passes over `bench.py --steps 3 --warmup 1 --streams 1 --batch 1`; "
       "fetch_x2 = FETCH_SIZE doubled per MI355X_MICROARCH.md 'HBM' (the counter tallies the 128-byte requests of wide coalesced streams at 64 bytes); WRITE_SIZE as reported",
       "kernels": {}}
for k, m in acc.items():
    n = max(cnt[k]["FETCH_SIZE"], 1)
    doc["kernels"][k] = {"calls": n, "fetch_raw": m["FETCH_SIZE"] / n * 1024.0, "fetch_x2": 2 * m["FETCH_SIZE"] / n * 1024.0, "write": m["WRITE_SIZE"] / max(cnt[k]["WRITE_SIZE"], 1) * 1024.0}
json.dump(doc, open(os.path.join(os.path.dirname(sys.argv[1]), "hbm_traffic.json"), "w"), indent=1)
print("%-46s %7s %14s %14s %14s" % ("kernel", "calls", "FETCH MB/call", "(x2) MB/call", "WRITE MB/call"))
for k, m in sorted(acc.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"]))[:14]:
    n = max(cnt[k]["FETCH_SIZE"], 1)
    print("%-46s %7d %14.2f %14.2f %14.2f" % (k, n, m["FETCH_SIZE"] / n / 1024.0, 2 * m["FETCH_SIZE"] / n / 1024.0, m["WRITE_SIZE"] / max(cnt[k]["WRITE_SIZE"], 1) / 1024.0))
PY
