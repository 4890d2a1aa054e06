#!/bin/bash
# Ablation builds of conv_wino.hip (WINO_ABL bit mask, see the source) linked against the instrumented objects of tools/build_trace.sh:
#   tools/wino_ablate.sh 0 63 8 ...   ->  tools/_trace/wino_abl_<mask>.so   (run with MOTIF_HIP_LIB=... python tools/trace_wino.py <shape>)
set -e
cd "$(dirname "$0")/../motif_amd/csrc"
[ -f /tmp/motif_trace_obj/api.o ] || bash ../../tools/build_trace.sh
for m in "$@"; do
  ( mkdir -p /tmp/wino_abl_$m
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -Wno-pass-failed -DMOTIF_TRACE -DWINO_ABL=$m $WINO_FLAGS -c conv_wino.hip -o /tmp/wino_abl_$m/conv_wino.o
    objs=$(ls /tmp/motif_trace_obj/*.o | grep -v conv_wino.o)
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_trace/wino_abl_$m.so $objs /tmp/wino_abl_$m/conv_wino.o ) &
done
wait
