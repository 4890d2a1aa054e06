#!/bin/bash
# build a variant of the library with extra compiler flags into tools/_trace/<name>.so:  tools/build_variant.sh name -DFOO=1 ...
set -e
name=$1; shift
cd "$(dirname "$0")/../motif_amd/csrc"
mkdir -p ../../tools/_trace /tmp/motif_var_$name
for f in api conv_igemm conv_split conv_split2 conv_wino conv_direct siren siren_split splat misc corr dcn; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -Wno-pass-failed "$@" -c $f.hip -o /tmp/motif_var_$name/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_trace/$name.so /tmp/motif_var_$name/*.o
