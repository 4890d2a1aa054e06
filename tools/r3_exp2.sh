#!/bin/bash
# round-3 experiment 2: which resource slows the compute phase?  (ablation builds: results are wrong, timing only)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
out=gpurun_out/r3/exp2_ablate.log
: > $out
for v in libmotif_hip pp_now pp_nob pp_nownob pp_nostage pp_nocommit; do
  echo "== variant $v" >> $out
  MOTIF_HIP_LIB=tools/_trace/$v.so timeout 300 python tools/trace_pp.py 6 2 2>&1 | grep -E "k=[1235] |block duration" >> $out
done
cat $out
