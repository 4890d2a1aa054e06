#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests -x -q -m gpu -k "dcn_concurrent or precontracted_stage or corrblock or conv or test_bench_two or dist" > gpurun_out/r3/verify_tests.log 2>&1; tail -4 gpurun_out/r3/verify_tests.log
timeout 900 python bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-roofline --verify-gather > gpurun_out/r3/bench_gloo2.json 2> gpurun_out/r3/bench_gloo2.err; tail -c 600 gpurun_out/r3/bench_gloo2.json; tail -3 gpurun_out/r3/bench_gloo2.err
timeout 1200 python bench.py --steps 10 --warmup 3 > gpurun_out/r3/bench_full.json 2> gpurun_out/r3/bench_full.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r3/bench_full.json") if l.startswith("{")][-1])
print("value %.1f M px/s  ms %.2f" % (d["value"]/1e6, d["ms_per_step"]))
print("roofline", {k:v for k,v in d["roofline"].items() if k in ("achieved","frac","overall")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cpu"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["sample"][-60:])
print("parity", d["parity"])
for k,v in d["stages"].items(): print("  %-16s %s" % (k, {kk:v[kk] for kk in v if kk in ("ms_per_clip","achieved","frac","calls","ms_instrumented","ms_in_stages")}))
PY
