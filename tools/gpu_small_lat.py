#!/usr/bin/env python3
"""Kernel time of 1 and of 4096 copies of the smallest reference streams: what a stream costs before its first command
(framing, header, prefix codes).  One-off, GPU box."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from brotli_rs_amd import brx
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import brx_knobs  # noqa: E402
G = os.path.join(ROOT, "tests", "golden", "data")
dev = torch.device("cuda:0")
ctx = brx_knobs.context(0)
for name in ("empty", "x", "10x10y", "64x", "quickfox", "ukkonooa", "backward65536", "quickfox_repeated", "monkey"):
    comp = open(os.path.join(G, name + ".compressed"), "rb").read()
    exp = open(os.path.join(G, name), "rb").read()
    cap = (len(exp) + 31) & ~15
    row = [name, len(comp), len(exp)]
    for n in (1, 4096):
        blob = torch.frombuffer(bytearray(comp * n), dtype=torch.uint8).to(dev)
        in_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * len(comp)
        out_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * cap
        out = torch.zeros(n * cap, dtype=torch.uint8, device=dev)
        out_len = torch.zeros(n, dtype=torch.int64, device=dev)
        status = torch.full((n,), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ts = []
        for _ in range(8):
            ctx.decode_batch_device(blob.data_ptr(), in_off.data_ptr(), n, out.data_ptr(), out_off.data_ptr(),
                                    out_len.data_ptr(), status.data_ptr(), timing=True)
            ctx.synchronize()
            ts.append(ctx.last_timing_ms(1))
        assert status.cpu().tolist() == [0] * n and out_len.cpu().tolist() == [len(exp)] * n
        row.append("%.1f us" % (1e3 * float(np.median(ts[2:]))))
    print(*row)
