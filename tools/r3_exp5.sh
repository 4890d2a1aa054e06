#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
out=gpurun_out/r3/exp5_trace.log; : > $out
for v in libmotif_hip pp_o3c1 pp_o1c0; do
  echo "== variant $v shape 6" >> $out
  MOTIF_HIP_LIB=tools/_trace/$v.so timeout 300 python tools/trace_pp.py 6 2>&1 | grep -E "k=[1234]|block duration|boundary|phase k=1" >> $out
done
cat $out
