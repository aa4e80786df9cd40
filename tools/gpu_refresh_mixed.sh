set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
BRX_FORCE_OVERLAP=1 timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
wl=mixed_textx4096
timeout 600 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --no-copy-path 2>/dev/null | tail -1 > $O/bench_r03_${wl}.json
WLS="mixed_textx4096 lcet10x4096 mapsdatazrhx4096 alice29x4096 backward65536x4096 quickfox_repeatedx8192 monkeyx16384 config5_1MiBx1024" bash tools/gpu_overlap_ab.sh
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python $R/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --no-copy-path --verify 0 > /tmp/kt.log 2>/dev/null
tail -1 /tmp/kt.log > $O/bench_r03_${wl}_under_rocprof.json
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r03_${wl}_kernel_stats.csv
cd $R; WL=mixed_textx4096 bash tools/gpu_overlap_trace.sh > /dev/null 2>&1
timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_r03_default.json
python3 -c "
import json; d=json.load(open('$O/bench_r03_default.json')); print(d['ms_per_step'], d['roofline']['traffic'], {k:v['frac_physical'] for k,v in d['copy_path'].items() if isinstance(v,dict)})
d=json.load(open('$O/bench_r03_mixed_textx4096.json')); print('mixed', d['roofline']['kernel_ms_avg'], d['roofline']['kernel_ms_median'])"
