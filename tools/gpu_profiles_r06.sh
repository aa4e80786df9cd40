#!/bin/bash
# Round 6's committed evidence (copied from gpurun_out/ into profiles/ by tools/collect_profiles.py r06): benches of every workload (with
# chain_floor), rocprofv3 kernel stats, SQ counters, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes), sweeps, the PCIe-inclusive
# host path, per-stream residency traces of the mixed batches, the default bench line.
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT; TAG=${TAG:-r06}
O=$R/gpurun_out; mkdir -p $O
cd $R
WLS=${WLS:-"alice29x4096 text40k_lowqx4096 text64k_q11x4096 raw_256KiBx4096 config5_1MiBx1024 gen_c5x1024 farcopy_1MiBx4096 backward65536x4096 quickfox_repeatedx8192 compressed_repeatedx4096 lcet10x4096 plrabn12x4096 mapsdatazrhx4096 mixed_textx4096 mixed_allx4096 monkeyx16384 ukkonooax16384 quickfoxx16384"}
for wl in $WLS; do
  timeout 600 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --no-copy-path 2>/dev/null | tail -1 > $O/bench_${TAG}_${wl}.json
done
echo "== sweeps (streams, MB/s, kernel ms)"
( for wl in alice29x4096 lcet10x4096 quickfox_repeatedx8192 backward65536x4096; do echo "== $wl"; WL=$wl NS="${NS:-1 256 1024 2048 4096 4352 8192 16384}" bash tools/gpu_sweep.sh 2>/dev/null; done ) | tee $O/${TAG}_sweep.txt
echo "== host path (PCIe-inclusive)"
( python tools/gpu_pcie.py 2>/dev/null; BRX_NO_MIRROR=1 python tools/gpu_pcie.py 2>/dev/null | head -1 ) | tee $O/${TAG}_pcie.txt
echo "== residency of the mixed batches (plan B)"
( for wl in mixed_textx4096 mixed_allx4096 mapsdatazrhx4096; do python tools/gpu_trace_streams.py $wl 1 2>&1 | grep -v amdgpu.ids; done ) | tee $O/${TAG}_residency.txt
cd /tmp && export TMPDIR=/tmp
for wl in $WLS; do
  rm -rf /tmp/kt
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python $R/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --no-copy-path --no-chain-floor --no-configs --verify 0 > /tmp/kt.log 2>/dev/null
  tail -1 /tmp/kt.log > $O/bench_${TAG}_${wl}_under_rocprof.json
  f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_${wl}_kernel_stats.csv
done
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
echo "== SQ counters, alice29 x 4096"
: > $O/${TAG}_pmc.txt
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/pmc_out
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_out -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-copy-path --no-chain-floor --no-configs --verify 0 > /dev/null 2>&1
  python3 - <<'PY' | tee -a $O/${TAG}_pmc.txt
import csv,glob,collections
agg=collections.defaultdict(float); disp=collections.defaultdict(set)
for f in glob.glob("/tmp/pmc_out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'brx' in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value'])
            if r['Kernel_Name'].startswith('brx_decode_kernel('): disp[r['Counter_Name']].add(r['Dispatch_Id'])
for k in sorted(agg): print("%-24s %18.0f per dispatch (%d dispatches)"%(k,agg[k]/max(1,len(disp[k])),len(disp[k])))
PY
done
cd $R
echo "== the default bench line (what the driver runs)"
timeout 900 python bench.py 2>/dev/null | tail -1 > $O/bench_${TAG}_default.json
python3 -c "import json;d=json.load(open('$O/bench_${TAG}_default.json'));print({k:d[k] for k in ('value','ms_per_step','bit_exact')}, d['roofline'], d.get('copy_path',{}).get('frac_physical'))"
