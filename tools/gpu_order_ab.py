"""Does the queue order matter for a heterogeneous batch of more streams than are resident?  Device path, caller order vs
BRX_OPT_ORDER (longest compressed stream first), kernel ms."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, brx_knobs
dev = torch.device("cuda:0")
for wl, n in (("mixed_textx4096", 8192), ("mixed_textx4096", 6144), ("mixed_allx4096", 8192)):
    names, _ = bench.WORKLOADS[wl]
    fx = [bench.load_fixture(f) for f in names]
    ctx = brx_knobs.context(0)
    b = bench.Batch(torch, np, dev, fx, n)
    for order in (False, True):
        ms = []
        for rep in range(5):
            ctx.decode_batch_device(b.blob.data_ptr(), b.in_off.data_ptr(), b.n, b.out.data_ptr(), b.out_off.data_ptr(), b.out_len.data_ptr(), b.status.data_ptr(), timing=True, order=order)
            ctx.synchronize()
            ms.append(ctx.last_timing_ms(1))
        print(wl, n, "order" if order else "caller order", "kernel ms", ["%.2f" % m for m in ms], "ok", b.verify(torch))
    ctx.close()
