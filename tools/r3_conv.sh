#!/bin/bash
# round-3 conv experiment: correctness, A/B against the two-block kernel (ENGINE=1), per-wave timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" > gpurun_out/r3/conv_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r3/conv_tests.log
out=gpurun_out/r3/conv_bench.log; : > $out
for cfg in "1 0" "0 0" "1 1" "0 1"; do
  set -- $cfg
  echo "== ENGINE=$1 RES=$2" >> $out
  if [ "$2" = "1" ]; then export RES=1; else unset RES; fi
  ENGINE=$1 REPS=30 timeout 300 python tools/conv_bench.py 2>&1 | grep shape >> $out
done
unset RES
out=gpurun_out/r3/conv_trace.log; : > $out
for sh in ${TRACE_SHAPES:-6 0 11}; do
  for v in ${TRACE_LIBS:-libmotif_hip}; do
    echo "== trace shape $sh lib $v" >> $out
    MOTIF_HIP_LIB=tools/_trace/$v.so timeout 300 python tools/trace_s2.py $sh 2>&1 | grep -v amdgpu.ids >> $out
  done
done
tail -3 gpurun_out/r3/conv_tests.log; cat gpurun_out/r3/conv_trace.log | head -${TRACE_LINES:-60}
paste <(grep -A18 "ENGINE=1 RES=0" gpurun_out/r3/conv_bench.log | cut -c1-75) <(grep -A18 "ENGINE=0 RES=0" gpurun_out/r3/conv_bench.log | cut -c43-75)
paste <(grep -A18 "ENGINE=1 RES=1" gpurun_out/r3/conv_bench.log | cut -c1-75) <(grep -A18 "ENGINE=0 RES=1" gpurun_out/r3/conv_bench.log | cut -c43-75)
