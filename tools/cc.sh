#!/bin/bash
# compile one kernel source for gfx950 and print register usage: tools/cc.sh siren_split [extra flags]
f=$1; shift
mkdir -p /tmp/t && cd /tmp/t && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I/root/repo/motif_amd/csrc -I/root/repo/include "$@" -c /root/repo/motif_amd/csrc/$f.hip -o /tmp/t/$f.o -save-temps=obj 2>&1 | grep -E "error|warning: v" | head -20
grep -E "^\s+\.name:|\.vgpr_count|vgpr_spill" /tmp/t/$f-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - | awk '{print $2,$4,$6}'
