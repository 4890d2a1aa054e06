#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" > gpurun_out/r3/exp3_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r3/exp3_tests.log
out=gpurun_out/r3/exp3_bench.log; : > $out
for cfg in "1 0 0" "0 0 0" "0 2 0" "0 3 0" "1 0 1" "0 0 1"; do
  set -- $cfg
  echo "== ENGINE=$1 RP=$2 RES=$3" >> $out
  if [ "$3" = "1" ]; then export RES=1; else unset RES; fi
  ENGINE=$1 RP=$2 REPS=30 timeout 300 python tools/conv_bench.py 2>&1 | grep shape >> $out
done
unset RES
out=gpurun_out/r3/exp3_trace.log; : > $out
for v in libmotif_hip pp_w2 pp_nt pp_w2nt; do
  echo "== variant $v shape 6 rp 2" >> $out
  MOTIF_HIP_LIB=tools/_trace/$v.so timeout 300 python tools/trace_pp.py 6 2 2>&1 | grep -E "k=[1234] |block duration|boundary" >> $out
done
for sh in "0 2" "0 3"; do
  set -- $sh
  echo "== trace shape $1 rp $2 RES" >> $out
  RES=1 MOTIF_HIP_LIB=tools/_trace/libmotif_hip.so timeout 300 python tools/trace_pp.py $1 $2 2>&1 | grep -E "k=|block duration|boundary" >> $out
done
tail -3 gpurun_out/r3/exp3_tests.log; cat gpurun_out/r3/exp3_trace.log
