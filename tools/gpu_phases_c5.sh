#!/bin/bash
# Where does a config-5 stream spend its time?  Phase timers (BRX_BRINGUP build) of one stream alone: the committed libbrotlienc fixture
# c5_0 next to a stream of the adaptive on-device generator (the hard variant: ~80 literal block switches, 16 meta-blocks, 2 trees).
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python - <<'PY'
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import brx_knobs, oracle_py
G = "tests/golden/data"
corpus = b"".join(open(os.path.join(G, t), "rb").read() for t in ("lcet10.txt", "alice29.txt", "plrabn12.txt", "asyoulik.txt"))
src = (corpus[(0 * 18211) % len(corpus):] + corpus)[:1 << 20]
ctx = brx_knobs.context(0)
st = ctx.generate_batch([src], metablock_bytes=65536, adaptive=True)[0]
open("/tmp/gen0.compressed", "wb").write(st)
for name, data in (("gen0", st), ("c5_0", open("tests/golden/config5/c5_0.compressed", "rb").read())):
    r = oracle_py.decode(data, want_stats=True)
    print(name, "compressed", len(data), "status", r[0], {k: r[2][k] for k in ("meta_blocks", "commands", "literals", "copies", "copy_bytes", "block_switches") if k in r[2]})
ctx.close()
PY
cp tests/golden/config5/c5_0.compressed /tmp/c5_0.compressed
BRX_BRINGUP=1 python brotli-rs_amd/build.py --force > /dev/null 2>&1
g++ -O2 -std=c++17 tools/diag_main.cpp -o tools/diag_main -Lbrotli-rs_amd -lbrx -Wl,-rpath,$GRAFT_REPO_ROOT/brotli-rs_amd -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 2>&1 | tail -3
for f in gen0 c5_0; do
  for n in 1 1024; do
    echo "== $f x $n"; BRX_DEBUG_STATS=1 timeout 120 ./tools/diag_main /tmp/$f.compressed 1048592 $n 3 2>&1 | grep -v amdgpu | grep "phases\|kernel\|stats" | tail -4
  done
done
python brotli-rs_amd/build.py --force > /dev/null 2>&1
