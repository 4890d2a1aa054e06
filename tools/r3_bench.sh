#!/bin/bash
# end-to-end A/B of the conv engines on the bench clip (c2)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for e in ${ENGINES:-1 0 2}; do
  MOTIF_CONV_ENGINE=$e timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg > gpurun_out/r3/bench_e$e.json 2> gpurun_out/r3/bench_e$e.err
  python - <<PY
import json
l=[x for x in open("gpurun_out/r3/bench_e$e.json") if x.startswith("{")]
d=json.loads(l[-1])
print("engine $e: value %.1f M px/s  ms/step %.2f  roofline %s" % (d["value"]/1e6, d["ms_per_step"], {k:d["roofline"][k] for k in ("achieved","frac")}))
st=d.get("stages",{})
for k,v in st.items():
    if isinstance(v,dict) and "ms" in v: print("   %-22s %6.2f ms  frac %s" % (k, v["ms"], v.get("frac")))
PY
done
