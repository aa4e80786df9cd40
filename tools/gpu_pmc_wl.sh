#!/bin/bash
# SQ instruction counters of one launch of a workload (WL=...), summed over the four kernel instances, per stream
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for wl in ${WLS:-backward65536x4096 quickfox_repeatedx8192}; do
  echo "== $wl"
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS"; do
    rm -rf /tmp/pmc_out
    timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_out -o p -- python $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --verify 0 > /dev/null 2>&1
    python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob("/tmp/pmc_out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'brx' in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=r['Kernel_Name'].startswith('brx_decode_kernel(')
for k in sorted(agg): print("%-24s %18.0f per launch"%(k,agg[k]/n[k]))
PY
  done
done
