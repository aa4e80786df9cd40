#!/bin/bash
# PC sampling of the decode kernel (rocprofv3 beta feature): where do the waves spend their issue slots.
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
rm -rf /tmp/pcs
timeout 120 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit ${UNIT:-time} --pc-sampling-method ${METHOD:-host_trap} --pc-sampling-interval ${INTERVAL:-1} --kernel-trace --output-format csv -d /tmp/pcs -o s -- $R/tools/diag_main $R/tests/golden/data/alice29.txt.compressed 152096 ${N:-1024} 2>&1 | tail -15
find /tmp/pcs -type f | head; 
f=$(find /tmp/pcs -name "*pc_sampling*csv" | head -1)
[ -n "$f" ] && { head -3 $f; wc -l $f; mkdir -p $R/gpurun_out; python3 - $f $R/gpurun_out/pcsamp.txt <<'PY'
import csv,sys,collections
c=collections.Counter()
rows=csv.DictReader(open(sys.argv[1]))
cols=rows.fieldnames
for r in rows:
    c[(r.get('Instruction') or r.get('Code_Object_Offset') or r.get('Instruction_Comment'), r.get('Code_Object_Offset'))]+=1
with open(sys.argv[2],'w') as o:
    o.write(str(cols)+"\n")
    for k,v in c.most_common(): o.write("%d\t%s\t%s\n"%(v,k[1],k[0]))
print(cols); print(c.most_common(5))
PY
}
