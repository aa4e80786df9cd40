"""debug: a few short streams through the C ABI, compared with the oracle (run on the GPU box)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py
from brotli_rs_amd import brx
import brx_knobs  # noqa: E402
g = os.path.join(ROOT, "tests", "golden")
names = [e["stream"] for e in json.load(open(os.path.join(g, "manifest.json")))]
streams = [open(os.path.join(g, "data", n), "rb").read() for n in names]
small = [(n, s) for n, s in zip(names, streams) if len(s) <= 500]
print(len(small), "small streams of", len(streams))
ctx = brx_knobs.context(0)
for n, s in small:
    want = oracle_py.decode(s, 0, cap=1 << 20)
    outs, status, out_len = ctx.decode_batch([s], 1 << 20)
    ok = int(status[0]) == want[0] and (want[0] != 0 or outs[0] == want[1])
    print("%-40s %4d B  oracle %2d  gpu %2d  len %7d  via_regular %d  %s" % (n, len(s), want[0], int(status[0]), int(out_len[0]), ctx.last_lean_listed(), "ok" if ok else "MISMATCH"))
    sys.stdout.flush()
ctx.close()
