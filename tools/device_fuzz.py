#!/usr/bin/env python3
"""Soak of the device-pointer path's work queue on a GPU box: ragged batches of MORE streams than workgroups (a small grid cap makes
every batch oversubscribed), in the caller's order -- the regular kernel then walks its queue big-first around the mean size the lean
kernel (plan A) or the pre-pass (plan B) summed, making only the walks that can have members.  Size mixes: all equal, two sizes, a few
huge among many tiny, everything between; outputs in unaligned slots; against the original bytes.  Usage: device_fuzz.py [rounds] [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import brotli_enc  # noqa: E402
import brx_knobs  # noqa: E402

G = os.path.join(ROOT, "tests", "golden", "data")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
corpus = b"".join(open(os.path.join(G, f), "rb").read() for f in ("lcet10.txt", "plrabn12.txt", "alice29.txt", "asyoulik.txt", "mapsdatazrh"))
dev = torch.device("cuda:0")
bad = 0
for r in range(rounds):
    cap_grid = rng.choice([16, 64, 256, 1024])
    levels = rng.choice([0, 1, 2])
    ctx = brx_knobs.context(0, grid_cap=cap_grid, levels=levels)
    kind = r % 5
    pool = []
    for _ in range(12):  # a dozen distinct streams per round, replicated
        if kind == 0:
            n = 30000
        elif kind == 1:
            n = rng.choice([2000, 200000])
        elif kind == 2:
            n = rng.choice([300] * 8 + [600000])
        elif kind == 3:
            n = int(2 ** rng.uniform(6, 19))
        else:
            n = rng.choice([50, 5000, 40000, 41000, 80000, 300000])
        o = rng.randrange(len(corpus) - n)
        data = corpus[o:o + n]
        q, lw = rng.choice([1, 5, 9, 11]), rng.randrange(16, 25)
        pool.append((brotli_enc.compress(data, quality=q, lgwin=lw), data, (q, lw)))
    nstreams = rng.choice([cap_grid + 1, 3 * cap_grid, 700, 2500])
    pick = [rng.choice(pool) if kind != 0 else pool[0] for _ in range(nstreams)]
    streams = [p[0] for p in pick]
    caps = [len(p[1]) + rng.randrange(0, 40) for p in pick]
    in_off = torch.tensor(np.concatenate([[0], np.cumsum([len(s) for s in streams])]), dtype=torch.int64, device=dev)
    out_off = torch.tensor(np.concatenate([[0], np.cumsum(caps)]), dtype=torch.int64, device=dev)
    blob = torch.frombuffer(bytearray(b"".join(streams)), dtype=torch.uint8).to(dev)
    out = torch.zeros(int(out_off[-1].item()) + 64, dtype=torch.uint8, device=dev)
    out_len = torch.zeros(nstreams, dtype=torch.int64, device=dev)
    status = torch.full((nstreams,), -1, dtype=torch.int32, device=dev)
    for rep in range(2):  # (the second launch may take the other plan)
        out.zero_(); status.fill_(-1)
        torch.cuda.synchronize()
        ctx.decode_batch_device(blob.data_ptr(), in_off.data_ptr(), nstreams, out.data_ptr(), out_off.data_ptr(), out_len.data_ptr(), status.data_ptr())
        ctx.synchronize()
        st = status.cpu().tolist(); ol = out_len.cpu().tolist(); host = out.cpu().numpy(); oo = out_off.cpu().tolist()
        for i, p in enumerate(pick):
            if st[i] != 0 or ol[i] != len(p[1]) or host[oo[i]:oo[i] + len(p[1])].tobytes() != p[1]:
                bad += 1
                if bad < 10:
                    print("MISMATCH round", r, "rep", rep, "stream", i, "status", st[i], "len", ol[i], len(p[1]), "quality / lgwin", p[2], "levels", levels,
                          "wide", [ctx.last_wide_streams(k) for k in (1, 2, 3)], "late", ctx.last_late_streams())
    print("round", r, "kind", kind, "grid cap", cap_grid, "levels", levels, "streams", nstreams, "done; mismatches so far", bad, flush=True)
    ctx.close()
sys.exit(1 if bad else 0)
