#!/bin/bash
# Round 6 A/B of the three small-stream changes (lane-parallel code-length code, no C++ call at a meta-block's end, periodic copies of
# 65..512 B inside the assembly loop): the library is rebuilt on the GPU box with each switched off (BRX_DEFS) and all on.
cd $GRAFT_REPO_ROOT
WLS=${WLS:-"flush1k_textx4096 flush1k_mixedx4096 monkeyx16384 alice29x4096 compressed_repeatedx4096"}
for defs in "BRX_NO_PAR_CLCODE BRX_NO_SKIP_END BRX_NO_PERIOD_COPY" "BRX_NO_PAR_CLCODE" "BRX_NO_SKIP_END" "BRX_NO_PERIOD_COPY" ""; do
  BRX_DEFS="$defs" python brotli-rs_amd/build.py --force > /dev/null 2>&1
  for wl in $WLS; do
    python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --no-copy-path --no-configs 2>/dev/null | tail -1 > /tmp/ab.json
    python -c "import json;d=json.load(open('/tmp/ab.json'));print('off=[%s] %-24s kernel %.4f ms  one stream %.4f ms  bit_exact %s' % ('$defs','$wl',d['roofline']['kernel_ms_avg'],d['roofline'].get('chain_floor_ms') or -1,d['bit_exact']))"
  done
done
