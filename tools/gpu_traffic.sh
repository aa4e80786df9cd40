#!/bin/bash
# HBM traffic per launch of brx_decode_kernel from the TCC counters, one counter per rocprofv3 pass (FETCH_SIZE and
# WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"), plus a --kernel-trace --stats pass.
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT; TAG=${TAG:-r01}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for wl in ${WORKLOADS:-alice29x4096 quickfox_repeatedx8192 backward65536x4096}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --verify 0 > /tmp/pmc_$c.log 2>&1
  done
  rm -rf /tmp/kt
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python $R/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --verify 0 > /tmp/kt.log 2>&1
  tail -1 /tmp/kt.log > $R/gpurun_out/bench_${TAG}_${wl}_under_rocprof.json
  f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/${TAG}_${wl}_kernel_stats.csv
  python3 - $wl $TAG <<'PY'
import csv,sys,glob,json,collections,os
wl,tag=sys.argv[1:3]; out={"workload":wl}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    fs=glob.glob("/tmp/pmc_%s/**/*counter_collection.csv"%c, recursive=True)
    vals=[]
    for f in fs:
        for r in csv.DictReader(open(f)):
            if 'brx_decode' in r['Kernel_Name'] and r['Counter_Name']==c: vals.append(float(r['Counter_Value']))
    out[c+"_per_dispatch_raw"]=vals
json.dump(out, open(os.environ['GRAFT_REPO_ROOT']+"/gpurun_out/traffic_%s_%s.json"%(tag,wl),"w"))
print(out)
PY
done
