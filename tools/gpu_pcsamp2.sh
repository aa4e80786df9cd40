#!/bin/bash
# PC sampling (rocprofv3 beta) of the decode kernel on a chosen stream: FILE CAP N REPS TAG.  Aggregated on the box.
set -u
R=$GRAFT_REPO_ROOT
FILE=$1; CAP=$2; N=$3; REPS=$4; TAG=$5
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
rm -rf /tmp/pcs_$TAG
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit ${UNIT:-time} --pc-sampling-method ${METHOD:-host_trap} --pc-sampling-interval ${INTERVAL:-1} --kernel-trace --output-format csv -d /tmp/pcs_$TAG -o s -- $R/tools/diag_main $R/$FILE $CAP $N $REPS 2>&1 | tail -6
f=$(find /tmp/pcs_$TAG -name "*pc_sampling*csv" | head -1)
echo "pc sampling file: $f"
[ -n "$f" ] && { head -3 $f; wc -l $f; mkdir -p $R/gpurun_out; python3 - $f $R/gpurun_out/pcsamp_$TAG.txt <<'PY'
import csv,sys,collections
c=collections.Counter()
rows=csv.DictReader(open(sys.argv[1]))
cols=rows.fieldnames
n=0
for r in rows:
    n+=1
    c[(r.get('Code_Object_Id'), r.get('Code_Object_Offset'), r.get('Instruction') or r.get('Instruction_Comment') or '')]+=1
with open(sys.argv[2],'w') as o:
    o.write(str(cols)+"\n%d samples\n"%n)
    for k,v in c.most_common(): o.write("%d\t%s\t%s\t%s\n"%(v,k[0],k[1],k[2]))
print(cols); print(n, c.most_common(5))
PY
}
