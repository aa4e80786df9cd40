#!/bin/bash
# Do the four instances of one plan-B launch really run at the same time?  Kernel start / end times from rocprofv3, per plan.
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-BRX_PLAN_B=1 BRX_PLAN_A=1}; do
  rm -rf /tmp/kt; echo "== variant: ${v:-default}  workload ${WL:-mixed_textx4096}"
  env $v timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o t -- python $R/bench.py --workload ${WL:-mixed_textx4096} --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-copy-path --no-chain-floor 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms_avg'], 'bit_exact', d['bit_exact'])"
  python3 - <<'PY'
import csv,glob
rows=[]
for f in glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'brx_decode' in r['Kernel_Name']: rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0], r.get('Queue_Id','?')))
rows.sort()
rows=rows[-(7 if len(rows) > 12 else 3):]
t0=rows[0][0] if rows else 0
for s,e,k,q in rows: print("%-24s queue %s start %10.3f ms  end %10.3f ms  (%8.3f ms)"%(k,q,(s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6))
PY
done 2>&1 | tee $R/gpurun_out/${TAG:-r04}_plan_trace_${WL:-mixed_textx4096}.txt
