#!/bin/bash
# build a variant of the library with extra compiler flags into tools/_trace/<name>.so:  tools/build_variant.sh name -DFOO=1 ...
# The source list is motif_amd/csrc/build.py's SOURCES (one list for every build script: a file added there is built here too).
set -e
name=$1; shift
cd "$(dirname "$0")/../motif_amd/csrc"
mkdir -p ../../tools/_trace /tmp/motif_var_$name
rm -f /tmp/motif_var_$name/*.o
for f in $(python3 -c "import build; print(' '.join(s[:-4] for s in build.SOURCES))"); do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -Wno-pass-failed "$@" -c $f.hip -o /tmp/motif_var_$name/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_trace/$name.so /tmp/motif_var_$name/*.o
