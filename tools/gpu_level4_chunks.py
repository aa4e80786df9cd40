#!/usr/bin/env python3
"""behind every batch launch sits one (usually empty) launch of the level-4 instance -- one workgroup per CU with 150
KiB of LDS, which can only be placed on a CU whose LDS is nearly free.  On the chunked host path (several HIP streams, the chunks'
launches next to each other) that empty launch of chunk k has to wait for room, and chunk k's output copy waits for it.  Measured
here: 4 x 4096 alice29 streams from / to pinned host buffers (4 chunks) and pageable ones, BRX_OPTION_LEVEL4 = 1 vs 0, best of 5."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from brotli_rs_amd import brx  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "data")
comp = open(os.path.join(GOLD, "alice29.txt.compressed"), "rb").read()
exp = open(os.path.join(GOLD, "alice29.txt"), "rb").read()
n = 4 * 4096
cap = (len(exp) + 15) & ~15
io = np.arange(n + 1, dtype=np.uint64) * len(comp)
oo = np.arange(n + 1, dtype=np.uint64) * cap
for pinned in (True, False):
    if pinned:
        hin, hout = brx.host_alloc(n * len(comp)), brx.host_alloc(n * cap)
    else:
        hin, hout = np.empty(n * len(comp), dtype=np.uint8), np.empty(n * cap, dtype=np.uint8)
    hin[:] = np.frombuffer(comp * n, dtype=np.uint8)
    for level4 in (1, 0, 1, 0):
        ctx = brx.Context(0, options={"level4": level4})
        best = 1e9
        for _ in range(6):
            t0 = time.perf_counter()
            st, ln = ctx.decode_batch_host_raw(hin.ctypes.data, io, n, hout.ctypes.data, oo)
            best = min(best, time.perf_counter() - t0)
        ok = (not st.any()) and hout[(n - 1) * cap:(n - 1) * cap + len(exp)].tobytes() == exp
        print("%s buffers, level4 launch %d: %d streams in %.2f ms (best of 6), bit-exact %s" % ("pinned" if pinned else "pageable", level4, n, best * 1e3, ok))
        ctx.close()
