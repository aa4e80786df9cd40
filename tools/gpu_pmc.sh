#!/bin/bash
set -u
G=$GRAFT_REPO_ROOT/tests/golden/data
cd /tmp && export TMPDIR=/tmp
N=${N:-4096}
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/pmc_out
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_out -o p -- $GRAFT_REPO_ROOT/tools/diag_main $G/alice29.txt.compressed 152096 $N > /dev/null 2>&1
  f=$(find /tmp/pmc_out -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if 'brx' in r['Kernel_Name']:
        agg[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for k in sorted(agg): print("%-24s %16.0f  per-wave %12.0f"%(k,agg[k]/n[k],agg[k]/n[k]/max(1,min(4096,int(__import__('os').environ.get('N','4096'))))))
PY
done
