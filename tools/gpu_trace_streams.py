"""Diagnostics: per-stream start / end / place of one launch (BRX_OPTION_TRACE), summarised per level and fixture."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, brx_knobs
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "mixed_textx4096"
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 2
names, n = bench.WORKLOADS[wl]
fx = [bench.load_fixture(f) for f in names]
ctx = brx_knobs.context(0, levels=levels, trace=1)
b = bench.Batch(torch, np, dev, fx, n)
for rep in range(3):
    b.step(ctx, timing=True); ctx.synchronize()
t = ctx.last_trace(n)
print(wl, "levels", levels, "kernel ms", ctx.last_timing_ms(1), "ok", b.verify(torch))
t0 = int(t[:, 0][t[:, 0] > 0].min())
K = len(fx)
for k in range(K):
    r = t[k::K]
    st = (r[:, 0].astype(np.int64) - t0) / 1e5; en = (r[:, 1].astype(np.int64) - t0) / 1e5
    lv = (r[:, 2] >> np.uint64(32)).astype(int)
    print("%-14s level %s  start ms: min %.2f median %.2f p90 %.2f max %.2f | end max %.2f | duration median %.2f" % (
        names[k], sorted(set(lv.tolist())), st.min(), np.median(st), np.percentile(st, 90), st.max(), en.max(), np.median(en - st)))
# residency: streams under way at a few instants, per level
st = (t[:, 0].astype(np.int64) - t0) / 1e5; en = (t[:, 1].astype(np.int64) - t0) / 1e5; lv = (t[:, 2] >> np.uint64(32)).astype(int)
for T in (0.5, 2, 5, 8, 12, 16, 20, 25, 30, 35, 40):
    on = (st <= T) & (en > T)
    print("t=%5.1f ms resident: " % T + "  ".join("L%d %4d" % (L, int((on & (lv == L)).sum())) for L in range(4)))
hw = t[:, 2] & np.uint64(0xffffffff)
# HW_ID gfx9: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx94x: se 14:13?), ... print distinct places early on
early = st <= 0.5
cu = ((hw >> np.uint64(8)) & np.uint64(0xf)).astype(int); se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(int)
print("streams started within 0.5 ms:", int(early.sum()), "by level", [int((early & (lv == L)).sum()) for L in range(4)])
ctx.close()
