#!/usr/bin/env python3
"""Differential fuzz of the node entry (brx_node_decode_batch) over VIRTUAL ranks on one GPU: random batches of the reference's data/
streams, encoder fixtures and their corrupted / truncated variants, random capacities (exact, too small, odd), random deal, number of
ranks and root, host pointers and device pointers (peer copies; every few rounds RCCL at one rank with the root's shard sent to
itself) -- status, length and the bytes of every slot must equal what ONE context gives for the same batch.
Usage: node_fuzz.py [rounds] [seed]"""
import glob
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from brotli_rs_amd import brx  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
G = os.path.join(ROOT, "tests", "golden")
pool = [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(G, "data", "*.compressed*")))]
pool += [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(G, "enc", "*.compressed")))[::3]]
pool = [p for p in pool if len(p) < 300000]
dev = torch.device("cuda:0")
ctx = brx.Context(0)
node = brx.Node([0] * 4)
rnode = brx.Node([0], options={"transport": 2, "exchange_root": 1})
bad = total = 0
for r in range(rounds):
    n = rng.choice((1, 2, 7, 64, 300, 1000))
    streams = []
    for _ in range(n):
        s = rng.choice(pool)
        k = rng.randrange(6)
        if k == 0 and len(s) > 2:
            s = s[:rng.randrange(1, len(s))]
        elif k == 1 and len(s) > 8:
            b = bytearray(s)
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
            s = bytes(b)
        streams.append(s)
    guess = [min(len(s) * rng.choice((2, 8, 40)), 1 << 20) + rng.randrange(0, 33) for s in streams]
    ref_out, ref_st, ref_len = ctx.decode_batch(streams, guess)  # (outputs only for status 0)
    # the slots as one context leaves them (bytes in front of an error included): decode once more into an arena we can look at
    in_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(s) for s in streams], out=in_off[1:])
    out_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(guess, out=out_off[1:])
    t_in = torch.frombuffer(bytearray(b"".join(streams) + b"\0"), dtype=torch.uint8).to(dev)
    t_io, t_oo = torch.from_numpy(in_off).to(dev), torch.from_numpy(out_off).to(dev)

    def run(fn):
        out = torch.full((int(out_off[-1]) + 16,), 0xEE, dtype=torch.uint8, device=dev)
        ln = torch.zeros(n, dtype=torch.int64, device=dev)
        st = torch.full((n,), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        fn(out, ln, st)
        torch.cuda.synchronize()
        return out.cpu().numpy(), st.cpu().numpy(), ln.cpu().numpy()

    want = run(lambda o, l, s: ctx.decode_batch_device(t_in.data_ptr(), t_io.data_ptr(), n, o.data_ptr(), t_oo.data_ptr(), l.data_ptr(), s.data_ptr()))
    deal = rng.choice(("ranges", "bytes", "snake"))
    ranks = rng.randrange(1, 5)
    root = rng.randrange(ranks)
    mode = rng.randrange(3)
    if mode == 0:
        what = "device, %d ranks, root %d, %s" % (ranks, root, deal)
        got = run(lambda o, l, s: node.decode_batch_device(t_in.data_ptr(), t_io.data_ptr(), n, o.data_ptr(), t_oo.data_ptr(), l.data_ptr(), s.data_ptr(),
                                                           deal=deal, use_gpus=ranks, root=root))
    elif mode == 1:
        what = "device, RCCL to itself, %s" % deal
        got = run(lambda o, l, s: rnode.decode_batch_device(t_in.data_ptr(), t_io.data_ptr(), n, o.data_ptr(), t_oo.data_ptr(), l.data_ptr(), s.data_ptr(),
                                                            deal=deal, use_gpus=1))
    else:
        what = "host, %d ranks, %s" % (ranks, deal)
        o_, st_, ln_ = node.decode_batch(streams, guess, deal=deal, use_gpus=ranks, raw=True)
        arena = np.full(int(out_off[-1]) + 16, 0xEE, dtype=np.uint8)
        for i, o in enumerate(o_):
            arena[int(out_off[i]):int(out_off[i]) + len(o)] = np.frombuffer(o, dtype=np.uint8)
        got = (arena, np.asarray(st_), np.asarray(ln_).astype(np.int64))
    total += 1
    ok = (got[1] == want[1]).all() and (got[2] == want[2]).all()
    why = ""
    if not ok:
        d = [i for i in range(n) if got[1][i] != want[1][i] or got[2][i] != want[2][i]][:5]
        why = "status / length differ at %s: got %s want %s (sizes %s caps %s)" % (d, [(int(got[1][i]), int(got[2][i])) for i in d],
                                                                                  [(int(want[1][i]), int(want[2][i])) for i in d], [len(streams[i]) for i in d], [guess[i] for i in d])
    if ok:
        for i in range(n):
            k = 0 if want[1][i] == 25 else min(int(want[2][i]), guess[i])  # what is in the slot: the stream, or the bytes in front of the
                                                                          # error (capacity too small: how far it got is not reported)
            a, b = int(out_off[i]), int(out_off[i]) + k
            if not (got[0][a:b] == want[0][a:b]).all():
                ok = False
                first = int(np.argmax(got[0][a:b] != want[0][a:b]))
                why = "bytes of stream %d differ from offset %d of %d (status %d, size %d, cap %d)" % (i, first, k, int(want[1][i]), len(streams[i]), guess[i])
                break
        if ok and mode != 2 and not (got[0][int(out_off[-1]):] == 0xEE).all():
            ok = False
            why = "bytes behind the last slot were written"
    if not ok:
        bad += 1
        print("MISMATCH round %d: %d streams, %s: %s" % (r, n, what, why))
    else:
        print("ok round %d: %d streams (%d fail), %s" % (r, n, int((want[1] != 0).sum()), what))
node.close()
rnode.close()
ctx.close()
print("node_fuzz seed %d: %d batches, %d mismatches" % (seed, total, bad))
sys.stdout.flush()
os._exit(1 if bad else 0)
