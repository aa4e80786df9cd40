#!/usr/bin/env python3
"""Repro of a node_fuzz finding: stream `idx` of round `rnd` of seed `seed`, decoded alone by ONE context at every input alignment
(the stream placed 0 .. 7 bytes into the input blob) and slot alignment, against the oracle."""
import glob, os, random, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py
from brotli_rs_amd import brx
seed, rnd, idx = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = random.Random(seed)
G = os.path.join(ROOT, "tests", "golden")
pool = [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(G, "data", "*.compressed*")))]
pool += [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(G, "enc", "*.compressed")))[::3]]
pool = [p for p in pool if len(p) < 300000]
for r in range(rnd + 1):
    n = rng.choice((1, 2, 7, 64, 300, 1000))
    streams = []
    for _ in range(n):
        s = rng.choice(pool)
        k = rng.randrange(6)
        if k == 0 and len(s) > 2:
            s = s[:rng.randrange(1, len(s))]
        elif k == 1 and len(s) > 8:
            b = bytearray(s); b[rng.randrange(len(b))] ^= 1 << rng.randrange(8); s = bytes(b)
        streams.append(s)
    guess = [min(len(s) * rng.choice((2, 8, 40)), 1 << 20) + rng.randrange(0, 33) for s in streams]
    deal = rng.choice(("ranges", "bytes", "snake")); ranks = rng.randrange(1, 5); root = rng.randrange(ranks); mode = rng.randrange(3)
s, cap = streams[idx], guess[idx]
open(os.path.join(ROOT, "gpurun_out", "nodefuzz_%d_%d_%d.compressed" % (seed, rnd, idx)), "wb").write(s)
st, want = oracle_py.decode(s, 0, cap=cap)[:2]
print("stream of %d B, cap %d: oracle status %d, %d bytes" % (len(s), cap, st, len(want)))
ctx = brx.Context(0)
for loop in (0,):
    ctx.set_option("command_loop", loop)
    for pad in (0,):
        for opad in (0, 1, 5, 8, 15):
            outs, status, ln = ctx.decode_batch([bytes(pad), s], [opad, cap])
            L = brx.load_library()
            # the slot's bytes whatever the status: decode again into an arena we can read
            blob = np.frombuffer(bytes(pad) + s, dtype=np.uint8).copy()
            in_off = np.array([0, pad, pad + len(s)], dtype=np.uint64); out_off = np.array([0, opad, opad + cap], dtype=np.uint64)
            out = np.full(opad + cap + 16, 0xEE, dtype=np.uint8)
            stt, lnn = ctx.decode_batch_host_raw(blob.ctypes.data, in_off, 2, out.ctypes.data, out_off)
            got = out[opad:opad + min(int(lnn[1]), cap)].tobytes()
            m = min(len(got), len(want))
            first = next((i for i in range(m) if got[i] != want[i]), None)
            print("loop %d input offset %d slot offset %d: status %d len %d, first difference to the oracle's prefix: %s" % (loop, pad, opad, int(stt[1]), int(lnn[1]), first))
            if first is not None and pad == 0:
                nd = sum(1 for i in range(m) if got[i] != want[i])
                last = max(i for i in range(m) if got[i] != want[i])
                print("   %d bytes differ, last at %d; got %s want %s" % (nd, last, got[first:first + 24].hex(), want[first:first + 24].hex()))
ctx.close()
