#!/bin/bash
# round-3 conv experiment: correctness, A/B of the engines (1 = two-block kernel, 2 / 3 / 4 = conv_split2 shapes) in isolation and in the model
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" 2>&1 | tail -2
out=gpurun_out/r3/conv_bench.log; : > $out
for e in ${ENGINES:-1 2 4}; do
  echo "== ENGINE=$e" >> $out
  ENGINE=$e REPS=20 timeout 300 python tools/conv_bench.py 2>&1 | grep shape >> $out
done
set -- ${ENGINES:-1 2 4}
paste <(grep -A18 "ENGINE=$1" $out | cut -c1-75) <(grep -A18 "ENGINE=$2" $out | cut -c43-75) <(grep -A18 "ENGINE=$3" $out | cut -c43-75)
A=${AB:-1 4}
for e in $A; do
MOTIF_BENCH_SHAPES=1 MOTIF_CONV_ENGINE=$e timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32-leg > /dev/null 2> gpurun_out/r3/shapes_e$e.txt
done
python - $A <<'PY'
import re, sys
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"# (\(.*?\))\s+(\d+)\s+([\d.]+)\s+([\d.]+)",l)
        if m: d[m.group(1)]=(int(m.group(2)),float(m.group(3)),float(m.group(4)))
    return d
e1,e2=sys.argv[1],sys.argv[2]
a=load("gpurun_out/r3/shapes_e%s.txt"%e1); b=load("gpurun_out/r3/shapes_e%s.txt"%e2)
tot1=tot2=0
for k,(n,ms,tf) in sorted(a.items(), key=lambda kv:-kv[1][1]):
    if k in b and ", 3, 3," in k:
        ms2=b[k][1]
        if abs(ms2-ms)/ms>0.02 and ms>0.1: print("%-38s x%-3d engine %s %6.3f ms  engine %s %6.3f ms  %+5.1f%%" % (k,n,e1,ms,e2,ms2,100*(ms2-ms)/ms))
        tot1+=ms; tot2+=ms2
print("total 3x3: engine %s %.2f  engine %s %.2f" % (e1,tot1,e2,tot2))
PY
