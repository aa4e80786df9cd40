"""Debug: the pulled reader on a generated text stream (meta-blocks of 2 MiB) under several input windows."""
import io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import brx_knobs, oracle_py
from brotli_rs_amd import brx
G = os.path.join(ROOT, "tests", "golden", "data")
rd = lambda n: open(os.path.join(G, n), "rb").read()
src = (rd("lcet10.txt") + rd("plrabn12.txt") + rd("alice29.txt")) * 6
mb = int(sys.argv[1]) if len(sys.argv) > 1 else (2 << 20)
c = brx_knobs.context(0)
st = c.generate_batch([src], metablock_bytes=mb, adaptive=True)[0]
r = oracle_py.decode(st, 0, cap=len(src) + 64)
print("stream", len(st), "oracle", r[0], r[1] == src)
for w in (8 << 20, 4 << 20, 2 << 20, 1 << 20):
    c.set_option("reader_window", w)
    before = c.stream_rollbacks()
    d = brx.Decompressor(io.BytesIO(st), c, streaming=True)
    got = bytearray(); err = None
    try:
        while True:
            ch = d.read(1 << 20)
            if not ch: break
            got += ch
    except ValueError as e:
        err = str(e)
    d.close()
    k = 0
    while k < min(len(got), len(src)) and got[k] == src[k]: k += 4096
    print("window", w >> 20, "MiB: got", len(got), "of", len(src), "prefix ok up to ~", k, "err", err, "rollbacks", c.stream_rollbacks() - before)
