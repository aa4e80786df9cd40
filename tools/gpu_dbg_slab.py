#!/usr/bin/env python3
"""Repro of the extended soak's one finding (device_fuzz 3 302, round 2): a few 600 KB quality-11 streams among hundreds of 300-byte
ones, 2500 streams on a grid capped at 64 waves -- which launch plan / level setting starves the slab pool?"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import brotli_enc, brx_knobs
G = os.path.join(ROOT, "tests", "golden", "data")
corpus = b"".join(open(os.path.join(G, f), "rb").read() for f in ("lcet10.txt", "plrabn12.txt", "alice29.txt", "asyoulik.txt", "mapsdatazrh"))
rng = random.Random(5)
pool = []
for k in range(12):
    n = 600000 if k % 4 == 0 else 300
    o = rng.randrange(len(corpus) - n)
    data = corpus[o:o + n]
    pool.append((brotli_enc.compress(data, quality=11, lgwin=22), data))
dev = torch.device("cuda:0")
for cap_grid in (64, 256, 0):
    for levels in (0, 1, 2):
        ctx = brx_knobs.context(0, grid_cap=cap_grid, levels=levels)
        pick = [rng.choice(pool) for _ in range(2500)]
        streams = [p[0] for p in pick]
        caps = [len(p[1]) + 7 for p in pick]
        in_off = torch.tensor(np.concatenate([[0], np.cumsum([len(s) for s in streams])]), dtype=torch.int64, device=dev)
        out_off = torch.tensor(np.concatenate([[0], np.cumsum(caps)]), dtype=torch.int64, device=dev)
        blob = torch.frombuffer(bytearray(b"".join(streams)), dtype=torch.uint8).to(dev)
        out = torch.zeros(int(out_off[-1].item()) + 64, dtype=torch.uint8, device=dev)
        out_len = torch.zeros(len(pick), dtype=torch.int64, device=dev)
        status = torch.full((len(pick),), -1, dtype=torch.int32, device=dev)
        for rep in range(2):
            status.fill_(-1); torch.cuda.synchronize()
            import time; t0 = time.time()
            ctx.decode_batch_device(blob.data_ptr(), in_off.data_ptr(), len(pick), out.data_ptr(), out_off.data_ptr(), out_len.data_ptr(), status.data_ptr())
            ctx.synchronize()
            st = status.cpu().tolist()
            print("grid cap", cap_grid, "levels", levels, "rep", rep, "bad", sum(1 for x in st if x != 0), "statuses", sorted(set(st)),
                  "wide", [ctx.last_wide_streams(k) for k in (1, 2, 3)], "late", ctx.last_late_streams(), "%.0f ms" % ((time.time() - t0) * 1e3), flush=True)
        ctx.close()
