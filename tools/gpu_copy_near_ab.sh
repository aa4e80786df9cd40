#!/bin/bash
# Round 6 A/B: ring-source copies of 513..8191 bytes inside the assembly loop (COPY_NEAR_MAX) against leaving for the C++ side at 512.
cd $GRAFT_REPO_ROOT
for defs in "COPY_NEAR_MAX=512" ""; do
  BRX_DEFS="$defs" python brotli-rs_amd/build.py --force > /dev/null 2>&1
  echo "=== [$defs]"
  python tools/gpu_fixture_rates.py 4096 2>/dev/null > /tmp/rates_$([ -z "$defs" ] && echo new || echo old).txt
  for wl in alice29x4096 compressed_repeatedx4096 monkeyx16384; do
    python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --no-copy-path --no-configs 2>/dev/null | tail -1 > /tmp/ab.json
    python -c "import json;d=json.load(open('/tmp/ab.json'));print('%-26s kernel %.4f ms exact %s' % ('$wl',d['roofline']['kernel_ms_avg'],d['bit_exact']))"
  done
done
python3 - <<'PY'
old={l.split()[2]:l for l in open('/tmp/rates_old.txt') if 'GB/s' in l}
new={l.split()[2]:l for l in open('/tmp/rates_new.txt') if 'GB/s' in l}
rows=[]
for k in old:
    if k in new:
        o=float(old[k].split(' ms')[0].split()[-1]); n=float(new[k].split(' ms')[0].split()[-1])
        rows.append((n/o,k,o,n,'NOT OK' in new[k]))
rows.sort()
print("fixture, ms at 512, ms at 8191, ratio (largest changes)")
for r in rows[:12]+rows[-5:]: print("%-18s %8.3f %8.3f  %.3f %s"%(r[1],r[2],r[3],r[0],"NOT OK" if r[4] else ""))
print("not ok:", sum(1 for r in rows if r[4]), "of", len(rows))
PY
