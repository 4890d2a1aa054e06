#!/bin/bash
# library variant with extra flags on conv_pp.hip only (other objects from motif_amd/csrc/build): tools/build_pp_variant.sh name -DFOO ...
set -e
name=$1; shift
cd "$(dirname "$0")/../motif_amd/csrc"
mkdir -p ../../tools/_trace
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -Wno-pass-failed -DMOTIF_TRACE "$@" -c conv_pp.hip -o /tmp/pp_$name.o
objs=$(ls build/*.o | grep -v conv_pp.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_trace/pp_$name.so $objs /tmp/pp_$name.o
