import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import brx_knobs, oracle_py
G = os.path.join(ROOT, "tests", "golden", "data")
rd = lambda n: open(os.path.join(G, n), "rb").read()
src = (rd("lcet10.txt") + rd("plrabn12.txt") + rd("alice29.txt")) * 6
c = brx_knobs.context(0)
for mb in (65536, 262144, 524288, 786432, 1 << 20, 1310720, 1572864, 2 << 20, 4 << 20):
    for adaptive in (True, False):
        st = c.generate_batch([src], metablock_bytes=mb, adaptive=adaptive)[0]
        r = oracle_py.decode(st, 0, cap=len(src) + 64, want_stats=True)
        print("mb", mb, "adaptive", adaptive, "stream", len(st), "oracle status", r[0], "ok", r[1] == src, "decoded", len(r[1]), "meta_blocks", r[2].get("meta_blocks"))
