"""Diagnostics: two launches from two contexts on two HIP streams at the same time -- how many waves per CU are resident together?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, brx_knobs
dev = torch.device("cuda:0")
specA, specB = sys.argv[1], sys.argv[2]   # e.g. alice29.txt:2048 lcet10.txt:1024
def mk(spec):
    name, n = spec.split(":"); n = int(n)
    ctx = brx_knobs.context(0, levels=0, trace=1)
    b = bench.Batch(torch, np, dev, [bench.load_fixture(name)], n)
    return ctx, b, n, name
A, B = mk(specA), mk(specB)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def go(X, s):
    ctx, b, n, _ = X
    ctx._lib  # noqa
    ctx.decode_batch_device(b.blob.data_ptr(), b.in_off.data_ptr(), n, b.out.data_ptr(), b.out_off.data_ptr(), b.out_len.data_ptr(), b.status.data_ptr(), hip_stream=s.cuda_stream)
for rep in range(2):
    go(A, sa); go(B, sb)
    torch.cuda.synchronize()
ta, tb = A[0].last_trace(A[2]), B[0].last_trace(B[2])
t0 = min(int(ta[:, 0].min()), int(tb[:, 0].min()))
for nm, t in ((A[3], ta), (B[3], tb)):
    st = (t[:, 0].astype(np.int64) - t0) / 1e5; en = (t[:, 1].astype(np.int64) - t0) / 1e5
    print("%-14s n %d level %s start median %.2f max %.2f end max %.2f duration median %.2f" % (nm, len(t), sorted(set((t[:, 2] >> np.uint64(32)).astype(int).tolist())), np.median(st), st.max(), en.max(), np.median(en - st)))
    for T in (0.3, 1, 3, 6, 10, 15, 20):
        print("   t=%5.1f resident %d" % (T, int(((st <= T) & (en > T)).sum())))
print("ok", A[1].verify(torch), B[1].verify(torch))
