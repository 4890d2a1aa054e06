#!/bin/bash
# splat A/B: kernel tests, then the bench clip's stage table (splat line) and value
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "splat or precontract or golden or parity" > gpurun_out/r3/splat_tests.log 2>&1; tail -3 gpurun_out/r3/splat_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg 2>/dev/null | tail -1 > gpurun_out/r3/bench_splat.json
python - <<PY
import json
d=json.load(open("gpurun_out/r3/bench_splat.json"))
print("value %.1f M px/s  ms/step %.2f" % (d["value"]/1e6, d["ms_per_step"]))
for k,v in d["stages"].items(): print("  %-14s %s" % (k, {a:b for a,b in v.items() if a in ("ms_per_clip","achieved","frac","ms_instrumented","ms_in_stages")}))
PY
