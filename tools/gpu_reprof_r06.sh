export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=r06
cd /tmp && export TMPDIR=/tmp
wl=alice29x4096
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python $R/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --no-copy-path --no-chain-floor --no-configs --verify 0 > /tmp/kt.log 2>/dev/null
tail -1 /tmp/kt.log > $O/bench_${TAG}_${wl}_under_rocprof.json
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_${wl}_kernel_stats.csv
head -3 $O/${TAG}_${wl}_kernel_stats.csv
# the default command the driver runs, under the profiler too (steps 20, warmup 5)
rm -rf /tmp/kt2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o t -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-copy-path --no-chain-floor > /tmp/kt2.log 2>/dev/null
tail -1 /tmp/kt2.log > $O/bench_${TAG}_default_under_rocprof.json
f=$(find /tmp/kt2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_default_kernel_stats.csv
head -3 $O/${TAG}_default_kernel_stats.csv
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
