#!/usr/bin/env python3
"""Incompressible payloads: libbrotlienc stores them as UNCOMPRESSED meta-blocks (the framing segment copies them through the ring).
How fast is a batch of those?  Usage: gpu_raw_bench.py [streams] [KiB per stream]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import brotli_enc, brx_knobs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
kib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rng = random.Random(3)
pool = []
for k in range(4):
    data = rng.randbytes(kib << 10)
    pool.append((brotli_enc.compress(data, quality=1, lgwin=22), data))
dev = torch.device("cuda:0")
ctx = brx_knobs.context(0)
pick = [pool[i % 4] for i in range(n)]
streams = [p[0] for p in pick]
caps = [(len(p[1]) + 15) & ~15 for p in pick]
in_off = torch.tensor(np.concatenate([[0], np.cumsum([len(s) for s in streams])]), dtype=torch.int64, device=dev)
out_off = torch.tensor(np.concatenate([[0], np.cumsum(caps)]), dtype=torch.int64, device=dev)
blob = torch.frombuffer(bytearray(b"".join(streams)), dtype=torch.uint8).to(dev)
out = torch.zeros(int(out_off[-1].item()) + 64, dtype=torch.uint8, device=dev)
out_len = torch.zeros(n, dtype=torch.int64, device=dev)
status = torch.full((n,), -1, dtype=torch.int32, device=dev)
best = 1e9
for rep in range(5):
    torch.cuda.synchronize()
    ctx.decode_batch_device(blob.data_ptr(), in_off.data_ptr(), n, out.data_ptr(), out_off.data_ptr(), out_len.data_ptr(), status.data_ptr(), timing=True)
    ctx.synchronize()
    best = min(best, ctx.last_timing_ms(1))
ok = bool((status == 0).all().item())
host = out.cpu().numpy(); oo = out_off.cpu().tolist()
ok = ok and all(host[oo[i]:oo[i] + len(pick[i][1])].tobytes() == pick[i][1] for i in range(0, n, max(1, n // 16)))
tot = sum(len(p[1]) for p in pick)
print("%d streams x %d KiB incompressible (compressed %d B): kernel %.3f ms = %.1f GB/s out (%.1f %% of 8 TB/s read + written), bit-exact %s"
      % (n, kib, len(pool[0][0]), best, tot / best / 1e6, 2 * tot / best / 1e6 / 8000 * 100, ok))
