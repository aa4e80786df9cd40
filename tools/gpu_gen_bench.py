#!/usr/bin/env python3
"""The on-device stream generator (brx_generate_batch) at batch scale: N x alice29.txt made into Brotli streams on the GPU,
compacted, decoded again on the GPU, compared -- nothing touches the host in between.  Prints both rates."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brotli_rs_amd import brx, shard  # noqa: E402
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import brx_knobs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
text = open('tests/golden/data/alice29.txt', 'rb').read()
dev = torch.device("cuda", 0)
ctx = brx_knobs.context(0)
one = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
blob = one.repeat(n).contiguous()
src_off = (torch.arange(n + 1, dtype=torch.int64, device=dev) * len(text)).contiguous()
slot = (ctx.generate_slot_bytes(len(text)) + 15) & ~15
comp = torch.zeros(n * slot, dtype=torch.uint8, device=dev)
comp_off = (torch.arange(n + 1, dtype=torch.int64, device=dev) * slot).contiguous()
comp_len = torch.zeros(n, dtype=torch.int64, device=dev)
st = torch.full((n,), -1, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    ctx.generate_batch_device(blob.data_ptr(), src_off.data_ptr(), n, comp.data_ptr(), comp_off.data_ptr(), comp_len.data_ptr(), st.data_ptr())
    best = min(best, time.perf_counter() - t0)
assert not st.any().item()
ratio = float(comp_len.sum().item()) / (n * len(text))
print("generate: %d x %d B -> %.1f %% of the input in %.1f ms (%.1f GB/s of input)" % (n, len(text), 100 * ratio, best * 1e3, n * len(text) / best / 1e9))
flat, offs = shard.compact(comp, comp_off, comp_len)
cap = (len(text) + 15) & ~15
out = torch.zeros(n * cap, dtype=torch.uint8, device=dev)
out_off = (torch.arange(n + 1, dtype=torch.int64, device=dev) * cap).contiguous()
out_len = torch.zeros(n, dtype=torch.int64, device=dev)
dst = torch.full((n,), -1, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
ms = []
for _ in range(5):
    ctx.decode_batch_device(flat.data_ptr(), offs.data_ptr(), n, out.data_ptr(), out_off.data_ptr(), out_len.data_ptr(), dst.data_ptr(), timing=True)
    ms.append(ctx.last_timing_ms(1))
ctx.synchronize()
assert not dst.any().item()
assert (out.view(n, cap)[:, :len(text)] == one[None, :]).all().item()
print("decode of the generated batch: kernel %.2f ms (%.1f GB/s decompressed), bit-exact" % (min(ms), n * len(text) / min(ms) / 1e6))
