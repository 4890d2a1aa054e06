#!/bin/bash
# counter passes (counters only) for splat_owner_kernel on tools/trace_splat.py; each --pmc set in its own run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
: > $O/pmc_splat.txt
export MOTIF_HIP_LIB=$R/motif_amd/libmotif_hip.so
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"; do
  d=$O/pmc_splat_tmp; rm -rf $d
  rocprofv3 --pmc $set -d $d -o t --output-format csv -- python3 $R/tools/trace_splat.py > /dev/null 2>$O/pmc_splat.err
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $O/pmc_splat.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "splat_owner" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
except Exception as e:
    print("no data", e)
for k, v in acc.items(): print("%-28s %16.0f per launch (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
  rm -rf $d
done
cat $O/pmc_splat.txt
