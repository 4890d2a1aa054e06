#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" 2>&1 | tail -2
for v in 1 0; do
MOTIF_BENCH_SHAPES=1 MOTIF_CONV_NOVEC=$v timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32-leg > gpurun_out/r3/igemm_v$v.json 2> gpurun_out/r3/shapes_v$v.txt
done
python - <<'PY'
import re, json
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"# (\(.*?\))\s+(\d+)\s+([\d.]+)\s+([\d.]+)",l)
        if m: d[m.group(1)]=(int(m.group(2)),float(m.group(3)),float(m.group(4)))
    return d
a=load("gpurun_out/r3/shapes_v1.txt"); b=load("gpurun_out/r3/shapes_v0.txt")
tot1=tot2=0
for k,(n,ms,tf) in sorted(a.items(), key=lambda kv:-kv[1][1]):
    if k in b:
        ms2=b[k][1]
        if abs(ms2-ms)/ms>0.03 and ms>0.04: print("%-38s x%-3d scalar %6.3f ms  vec %6.3f ms  %+5.1f%%" % (k,n,ms,ms2,100*(ms2-ms)/ms))
        tot1+=ms; tot2+=ms2
print("total conv: scalar-staging %.2f  vec-staging %.2f" % (tot1,tot2))
for v in (1,0):
    d=json.loads([l for l in open("gpurun_out/r3/igemm_v%d.json"%v) if l.startswith("{")][-1])
    print("novec=%d: %.1f M px/s  conv_other %.3f ms" % (v, d["value"]/1e6, d["stages"]["conv_other"]["ms_per_clip"]))
PY
