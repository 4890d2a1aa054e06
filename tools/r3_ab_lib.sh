#!/bin/bash
# A/B of two library builds on the conv microbench (ENGINE as given) and on the bench clip: tools/r3_ab_lib.sh <variant.so name in tools/_trace>
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
V=$GRAFT_REPO_ROOT/tools/_trace/$1.so
for rep in 1 2; do
for lib in default $V; do
  if [ $lib = default ]; then unset MOTIF_HIP_LIB; else export MOTIF_HIP_LIB=$lib; fi
  echo "== $lib  ENGINE=${ENGINE:-2}"
  ENGINE=${ENGINE:-2} REPS=20 timeout 300 python tools/conv_bench.py 2>&1 | grep shape | head -${NSHAPES:-8}
done
done
for lib in default $V default $V; do
  if [ $lib = default ]; then unset MOTIF_HIP_LIB; else export MOTIF_HIP_LIB=$lib; fi
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'value %.1f M px/s' % (d['value']/1e6), 'conv3x3 %.2f ms' % d['stages']['conv3x3']['ms_per_clip'])"
done
