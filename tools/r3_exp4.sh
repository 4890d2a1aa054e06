#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" > gpurun_out/r3/exp4_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r3/exp4_tests.log
out=gpurun_out/r3/exp4_bench.log; : > $out
for cfg in "1 0" "0 0" "1 1" "0 1"; do
  set -- $cfg
  echo "== ENGINE=$1 RES=$2" >> $out
  if [ "$2" = "1" ]; then export RES=1; else unset RES; fi
  ENGINE=$1 REPS=30 timeout 300 python tools/conv_bench.py 2>&1 | grep shape >> $out
done
unset RES
out=gpurun_out/r3/exp4_trace.log; : > $out
for sh in "6 0" "0 0" "0 1" "11 0"; do
  set -- $sh
  echo "== trace shape $1 RES=$2" >> $out
  if [ "$2" = "1" ]; then export RES=1; else unset RES; fi
  MOTIF_HIP_LIB=tools/_trace/libmotif_hip.so timeout 300 python tools/trace_pp.py $1 2>&1 | grep -E "k=|block duration|boundary" >> $out
done
tail -3 gpurun_out/r3/exp4_tests.log; cat gpurun_out/r3/exp4_trace.log | head -70; paste <(grep -A18 "ENGINE=1 RES=0" $GRAFT_REPO_ROOT/gpurun_out/r3/exp4_bench.log | cut -c1-75) <(grep -A18 "ENGINE=0 RES=0" $GRAFT_REPO_ROOT/gpurun_out/r3/exp4_bench.log | cut -c45-75)
