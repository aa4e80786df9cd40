#!/bin/bash
# usage: F=<fixture> CAP=<cap> bash tools/gpu_pmc1.sh  -> per-wave instruction counts for one stream
set -u
G=$GRAFT_REPO_ROOT/tests/golden/data
cd /tmp && export TMPDIR=/tmp
for F in ${FILES}; do
rm -rf /tmp/pmc_out
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY --output-format csv -d /tmp/pmc_out -o p -- $GRAFT_REPO_ROOT/tools/diag_main $G/$F 1000000 1 > /dev/null 2>&1
f=$(find /tmp/pmc_out -name "*counter_collection.csv" | head -1)
echo "== $F"
python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if 'brx' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])
print("  ".join("%s=%d"%(k.replace('SQ_',''),v) for k,v in sorted(agg.items())))
PY
done
