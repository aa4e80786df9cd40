#!/usr/bin/env python3
"""Host-pointer path, PCIe-inclusive: 4096 x alice29 from host buffers (pinned via brx_host_alloc / pageable numpy) through
brx_decode_batch -- copy in, decode, output back -- wall time of the call.  With pinned output the kernel stores the output
to host memory itself while it decodes (BrxKernelArgs::out_mirror); BRX_NO_MIRROR=1 copies it back afterwards instead."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brotli_rs_amd import brx  # noqa: E402
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import brx_knobs  # noqa: E402

comp = open('tests/golden/data/alice29.txt.compressed', 'rb').read()
exp = open('tests/golden/data/alice29.txt', 'rb').read()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cap = (len(exp) + 15) & ~15
ctx = brx_knobs.context(0)
tag = "output copied back after the decode (BRX_NO_MIRROR)" if os.environ.get("BRX_NO_MIRROR") else "output stored to host memory by the kernel"
for kind in ('pinned', 'pageable'):
    if kind == 'pinned':
        a, b = brx.host_alloc(len(comp) * n), brx.host_alloc(cap * n)
    else:
        a, b = np.zeros(len(comp) * n, dtype=np.uint8), np.zeros(cap * n, dtype=np.uint8)
    a[:] = np.frombuffer(comp * n, dtype=np.uint8)
    io = np.arange(n + 1, dtype=np.uint64) * len(comp)
    oo = np.arange(n + 1, dtype=np.uint64) * cap
    best = 1e9
    for r in range(5):
        b[:] = 0
        t0 = time.perf_counter()
        st, ln = ctx.decode_batch_host_raw(a.ctypes.data, io, n, b.ctypes.data, oo)
        best = min(best, time.perf_counter() - t0)
    assert not st.any() and all(int(x) == len(exp) for x in ln)
    want = np.frombuffer(exp, dtype=np.uint8)
    got = b.reshape(n, cap)[:, :len(exp)]
    assert (got == want[None, :]).all()
    print("%s host buffers%s: %d x alice29, H2D + decode + output on the host: %.2f ms wall, %.1f GB/s decompressed (PCIe-inclusive)"
          % (kind, " (" + tag + ")" if kind == 'pinned' else "", n, best * 1e3, n * len(exp) / best / 1e9))
