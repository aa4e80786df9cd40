"""Debug: which level decodes what, per launch plan (device path, like bench.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, brx_knobs
dev = torch.device("cuda:0")
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["alice29.txt", "asyoulik.txt", "plrabn12.txt", "lcet10.txt"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
fx = [bench.load_fixture(f) for f in names]
for levels in (0, 2, 1):
    ctx = brx_knobs.context(0, levels=levels)
    b = bench.Batch(torch, np, dev, fx, n)
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.step(ctx, timing=True)
        ctx.synchronize(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("levels=%d rep %d: wall %.3f ms kernel %.3f ms  wide>=1 %d >=2 %d >=3 %d late %d redo %d lean_listed %d ok %s" % (
            levels, rep, dt * 1e3, ctx.last_timing_ms(1), ctx.last_wide_streams(1), ctx.last_wide_streams(2), ctx.last_wide_streams(3),
            ctx.last_late_streams(), ctx.last_redo_bytes(), ctx.last_lean_listed(), b.verify(torch)))
    ctx.close()
