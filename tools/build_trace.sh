#!/bin/bash
# instrumented build (-DMOTIF_TRACE -DMOTIF_SIREN_DBG) of the library into tools/_trace/ (git-ignored); use with MOTIF_HIP_LIB
set -e
cd "$(dirname "$0")/../motif_amd/csrc"
mkdir -p ../../tools/_trace /tmp/motif_trace_obj
for f in $(python3 -c "import build; print(' '.join(s[:-4] for s in build.SOURCES))"); do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -Wno-pass-failed -DMOTIF_TRACE -DMOTIF_SIREN_DBG -c $f.hip -o /tmp/motif_trace_obj/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_trace/libmotif_hip.so /tmp/motif_trace_obj/*.o
