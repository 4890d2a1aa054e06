#!/bin/bash
# round-3 experiment 1: correctness of the ping-pong conv kernel, A/B against the two-block kernel, phase trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" > gpurun_out/r3/exp1_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r3/exp1_tests.log
for cfg in "1 0" "0 0" "0 2" "0 3"; do
  set -- $cfg
  echo "== ENGINE=$1 RP=$2" >> gpurun_out/r3/exp1_bench.log
  ENGINE=$1 RP=$2 REPS=30 timeout 300 python tools/conv_bench.py >> gpurun_out/r3/exp1_bench.log 2>&1
done
for sh in "0 2" "0 3" "6 2" "11 2"; do
  set -- $sh
  echo "== trace shape $1 rp $2" >> gpurun_out/r3/exp1_trace.log
  MOTIF_HIP_LIB=tools/_trace/libmotif_hip.so timeout 300 python tools/trace_pp.py $1 $2 >> gpurun_out/r3/exp1_trace.log 2>&1
done
tail -5 gpurun_out/r3/exp1_tests.log
