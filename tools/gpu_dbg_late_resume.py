import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py, brx_knobs
s = open(os.path.join(ROOT, "tests/golden/regress_late/r04_wide43_1_90.compressed"), "rb").read()
st, exp = oracle_py.decode(s, cap=1 << 20)[:2]
ctx = brx_knobs.context(0, levels=2)  # plan B on every launch
for pad in (0, 1, 2, 3, 4, 8, 12, 16, 17):
    caps = [len(exp) + pad] * 2
    outs, status, out_len = ctx.decode_batch([s, s], caps)
    for i, (o, t) in enumerate(zip(outs, status)):
        if int(t) != 0 or o != exp:
            k = next((j for j in range(min(len(o), len(exp))) if o[j] != exp[j]), -1)
            ndiff = sum(1 for x, y in zip(o, exp) if x != y)
            # where does the wrong run come from?
            seg = o[k:k + 12]
            at = exp.find(seg, max(0, k - 300000), k + 12)
            print("pad", pad, "stream", i, "skew", (i * caps[0]) % 16, "BAD st", int(t), "first diff", k, "bytes differing", ndiff, "got", seg, "exp", exp[k:k + 12], "got-run found in exp at", at, "delta", k - at)
        else:
            print("pad", pad, "stream", i, "skew", (i * caps[0]) % 16, "ok", "late", ctx.last_late_streams(), "wide", [ctx.last_wide_streams(q) for q in (1, 2, 3)])
ctx.close()
