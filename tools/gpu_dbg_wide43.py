"""Bring-up: the stream of tools/wide_fuzz.py 3 43 (round 1, stream 90) that decoded to wrong bytes with status 0 under BRX_GRID_CAP=64."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py, brx_knobs
s = open(os.path.join(ROOT, "tests/golden/regress_late/r04_wide43_1_90.compressed"), "rb").read()
st, exp = oracle_py.decode(s, cap=1 << 20)[:2]
assert st == 0
small = open(os.path.join(ROOT, "tests/golden/data/alice29.txt.compressed"), "rb").read()
for label, kw in (("default", {}), ("levels=1", dict(levels=1)), ("loop_build=0", dict(loop_build=0)), ("loop_build=1", dict(loop_build=1)), ("command_loop=1 (C++ only)", dict(command_loop=1))):
    try:
        ctx = brx_knobs.context(0, **kw)
    except Exception as e:
        print(label, "ctx failed", e); continue
    for batch in ([s], [s] * 3, [small] * 5 + [s] + [small] * 5, [s] * 70):
        outs, status, out_len = ctx.decode_batch(batch, len(exp) + 17)
        res = []
        for b, o, t in zip(batch, outs, status):
            if b is s:
                if int(t) != 0 or o != exp:
                    k = next((i for i in range(min(len(o), len(exp))) if o[i] != exp[i]), min(len(o), len(exp)))
                    res.append("BAD st %d len %d first diff at %d" % (int(t), len(o), k))
                else:
                    res.append("ok")
        print(label, "batch of", len(batch), "->", sorted(set(res)), "wide", [ctx.last_wide_streams(k) for k in (1, 2, 3)], flush=True)
    ctx.close()
