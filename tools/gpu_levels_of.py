"""Which kernel level decodes each fixture (tables vs LDS table memory)?  One stream per launch, plan A."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import brx_knobs
g = os.path.join(ROOT, "tests", "golden")
files = [os.path.join(g, "data", e["stream"]) for e in json.load(open(os.path.join(g, "manifest.json")))]
files += [os.path.join(g, "enc", e["name"] + ".compressed") for e in json.load(open(os.path.join(g, "enc", "manifest.json")))["streams"]]
files += [os.path.join(g, "config5", "c5_%d.compressed" % k) for k in range(4)]
ctx = brx_knobs.context(0, levels=0)
by = {0: [], 1: [], 2: [], 3: []}
for f in files:
    d = open(f, "rb").read()
    outs, st, ln = ctx.decode_batch([d], 1 << 21)
    lvl = 3 if ctx.last_wide_streams(3) else 2 if ctx.last_wide_streams(2) else 1 if ctx.last_wide_streams(1) else 0
    by[lvl].append((os.path.basename(f), len(d), int(st[0]), ctx.last_late_streams()))
for l in range(4):
    print("level", l, len(by[l]), "streams")
    if l:
        for x in by[l]: print("   ", x)
ctx.close()
