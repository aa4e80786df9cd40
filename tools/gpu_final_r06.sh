#!/bin/bash
# Round 6, final kernel (after the ring reload fix changed the kernel source id): the default bench line as the driver runs it, the
# rocprofv3 kernel stats of the headline and of the default command, HBM traffic passes (FETCH_SIZE / WRITE_SIZE apart) for the
# workloads of profiles/hbm_traffic.json, smoke.  tools/collect_profiles.py r06 copies the results into profiles/.
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=r06; mkdir -p $O
cd $R
( time python bench.py ) > $O/bench_default_final.log 2>&1; tail -4 $O/bench_default_final.log | head -1 | cut -c1-400
grep '^{' $O/bench_default_final.log | tail -1 > $O/bench_${TAG}_default_final.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for wl in config5_1MiBx1024 text64k_q11x4096 mixed_allx4096; do
  timeout 600 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --no-copy-path --no-configs 2>/dev/null | tail -1 > $O/bench_${TAG}_${wl}.json
done
cd /tmp && export TMPDIR=/tmp
for wl in alice29x4096 config5_1MiBx1024; do
  rm -rf /tmp/kt
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python $R/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --no-copy-path --no-chain-floor --no-configs --verify 0 > /tmp/kt.log 2>/dev/null
  tail -1 /tmp/kt.log > $O/bench_${TAG}_${wl}_under_rocprof.json
  f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_${wl}_kernel_stats.csv
  head -2 $O/${TAG}_${wl}_kernel_stats.csv
done
rm -rf /tmp/kt2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o t -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-copy-path --no-chain-floor > /tmp/kt2.log 2>/dev/null
tail -1 /tmp/kt2.log > $O/bench_${TAG}_default_under_rocprof.json
f=$(find /tmp/kt2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_default_kernel_stats.csv
for wl in alice29x4096 config5_1MiBx1024 gen_c5x1024 lcet10x4096 farcopy_1MiBx4096 backward65536x4096 quickfox_repeatedx8192; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-copy-path --no-chain-floor --no-configs --verify 0 > /dev/null 2>&1
  done
  python3 - $wl $TAG <<'PY'
import csv,sys,glob,json,os
wl,tag=sys.argv[1:3]; out={"workload":wl}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    tot=0.0; launches=set()   # one launch = the regular kernel + the other instances around it
    for f in glob.glob("/tmp/pmc_%s/**/*counter_collection.csv"%c, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'brx_decode' in r['Kernel_Name'] and r['Counter_Name']==c:
                tot+=float(r['Counter_Value'])
                if r['Kernel_Name'].startswith('brx_decode_kernel('): launches.add(r['Dispatch_Id'])
    out[c+"_per_dispatch_raw"]=[tot/len(launches)] if launches else []
json.dump(out, open(os.environ['GRAFT_REPO_ROOT']+"/gpurun_out/traffic_%s_%s.json"%(tag,wl),"w"))
print(wl, {k:(sum(v)/max(1,len(v)) if isinstance(v,list) else v) for k,v in out.items()})
PY
done
