"""Timing of the truncation sweep pieces (why did the test take 25 minutes?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import brx_knobs, oracle_py as oracle, craft
G = os.path.join(ROOT, "tests", "golden")
rd = lambda n: open(os.path.join(G, "data", n), "rb").read()
c5 = open(os.path.join(G, "config5", "c5_0.compressed"), "rb").read()
srcs = [("monkey", rd("monkey.compressed"), 10 ** 6), ("l1", craft.growing_tables_stream(81, [105], mode=2, n_cmds=60)[0], 10 ** 6), ("c5_0", c5, 1200), ("alice29", rd("alice29.txt.compressed"), 1500), ("lcet10", rd("lcet10.txt.compressed"), 2600), ("maps", rd("mapsdatazrh.compressed"), 3600),
        ("g150_250", craft.growing_tables_stream(82, [3, 150, 2, 250], mode=1, n_cmds=20)[0], 10 ** 6), ("g256", craft.growing_tables_stream(83, [256], mode=3, n_cmds=12)[0], 10 ** 6)]
ctx = brx_knobs.context(0, levels=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for name, data, upto in srcs:
    cuts = []
    for k in range(1, min(len(data), upto) + 1):
        cuts.append(data[:k])
        if k % 4 == 0:
            for j in (1, 3, 6):
                cuts.append(data[:k - 1] + bytes([data[k - 1] & ((1 << j) - 1)]))
    t0 = time.time(); want = [oracle.decode(s_, 0, cap=1 << 16) for s_ in cuts]; t1 = time.time()
    outs, status, out_len = ctx.decode_batch(cuts, 1 << 16); t2 = time.time()
    bad = sum(1 for w, st in zip(want, status) if w[0] != st)
    print("%-10s %5d cuts: oracle %.2f s, gpu %.2f s, kernel %.2f ms, mismatches %d, wide %d late %d" % (name, len(cuts), t1 - t0, t2 - t1, ctx.last_timing_ms(1), bad, ctx.last_wide_streams(1), ctx.last_late_streams()), flush=True)
