#!/usr/bin/env python3
"""Differential fuzz of what a slot holds when its stream FAILS, against the oracle: random batches of the reference's data/ streams,
encoder fixtures and their truncated / bit-flipped variants in slots of random odd capacities (so every slot alignment mod 16 turns
up, and capacities that are exact, too small, or far too big), one context, random command loop and launch plan.  For every stream:
status equals the oracle's; for status 0 length and bytes; for a failing stream (truncated: 24; or any format error) the bytes in FRONT of the error:
the slot and the oracle's output agree over the shorter of the two lengths.  (How MANY bytes are out when the input ends inside a
command is not defined by the reference either: src/lib.rs:2173-2193 hands decompress() the caller's buffer and drops what the failing
call had written into it, so the delivered prefix depends on the caller's buffer size; out_len is specified for status 0 and 25 only,
brx.h.  A truncated stream in a slot smaller than the UNCUT stream's output may also end with 25: a command that announces more
output than fits is refused before its input is read.)  Nothing is written behind a slot's capacity.
(tools/small_fuzz.py / wide_fuzz.py compare status and the bytes of streams that decode; tools/node_fuzz.py compares the node with
one context -- round 6's seg_resume bug was two GPU paths disagreeing, this fuzz is the one that asks the oracle.)
Usage: prefix_fuzz.py [rounds] [seed]"""
import glob
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py  # noqa: E402
from brotli_rs_amd import brx  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
G = os.path.join(ROOT, "tests", "golden")
pool = [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(G, "data", "*.compressed*")))]
pool += [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(G, "enc", "*.compressed")))[::2]]
pool = [p for p in pool if len(p) < 200000]
memo = {}


def oracle(s):
    if s not in memo:
        memo[s] = oracle_py.decode(s, 0, cap=4 << 20)[:2]
    return memo[s]


bad = total = checked_prefix = 0
for r in range(rounds):
    n = rng.choice((3, 16, 100, 400))
    loop = rng.choice((0, 0, 0, 6, 8))
    plan = rng.choice(("", "", "plan_a", "plan_b"))
    opts = {"command_loop": loop}
    if plan:
        opts["levels"] = 0 if plan == "plan_a" else 2
    if rng.randrange(4) == 0:
        opts["grid_cap"] = rng.choice((8, 64))
    ctx = brx.Context(0, options=opts)
    streams, caps, want = [], [], []
    for _ in range(n):
        s = s0 = rng.choice(pool)
        k = rng.randrange(5)
        if k <= 1 and len(s) > 2:
            s = s[:rng.randrange(1, len(s))]
        elif k == 2 and len(s) > 8:
            b = bytearray(s)
            for _ in range(rng.randrange(1, 3)):
                b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
            s = bytes(b)
        st, out = oracle(s)
        if st not in (0, 24) and len(out) > (3 << 20):
            continue
        full = len(oracle(s0)[1]) if (k <= 1 and oracle(s0)[0] == 0) else None  # the uncut stream's size, when there is one
        c = rng.randrange(4)
        cap = len(out) + (0 if c == 0 else rng.randrange(1, 40) if c == 1 else rng.randrange(1, 70000) if c == 2 else -rng.randrange(0, len(out) + 1))
        streams.append(s)
        caps.append(max(cap, 0))
        want.append((st, out, full))
    n = len(streams)
    in_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(s) for s in streams], out=in_off[1:])
    out_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(caps, out=out_off[1:])
    blob = np.frombuffer(b"".join(streams) + b"\0" * 16, dtype=np.uint8).copy()
    arena = np.full(int(out_off[-1]) + 64, 0xEE, dtype=np.uint8)
    status, out_len = ctx.decode_batch_host_raw(blob.ctypes.data, in_off, n, arena.ctypes.data, out_off)
    ctx.close()
    total += n
    nb = 0
    for i in range(n):
        st, out, full = want[i]
        a, cap = int(out_off[i]), caps[i]
        got_st, got_len = int(status[i]), int(out_len[i])
        ok, why = True, ""
        if st == 0:
            if len(out) > cap:
                ok, why = got_st == 25, "status %d, want 25 (needs %d, capacity %d)" % (got_st, len(out), cap)
            else:
                ok = got_st == 0 and got_len == len(out) and arena[a:a + len(out)].tobytes() == out
                why = "status %d length %d, want 0 and %d, or bytes differ (slot offset %d mod 16)" % (got_st, got_len, len(out), a % 16)
        elif got_st == 25 and (st != 24 or full is None or cap < full):
            pass  # a failing stream may run out of room before it runs into its error
        elif got_st != st:
            ok, why = False, "status %d want %d (capacity %d, the oracle got to %d)" % (got_st, st, cap, len(out))
        else:  # (24 = truncated; the other kinds fail at a defined bit too, and what was decoded before it is the same on both sides)
            m = min(got_len, len(out), cap)
            ok = got_len <= cap and arena[a:a + m].tobytes() == out[:m]
            why = "length %d beyond the capacity %d" % (got_len, cap) if got_len > cap else "failing stream (status %d): bytes differ from %d of %d (slot offset %d mod 16)" % (
                st, next((j for j in range(m) if arena[a + j] != out[j]), -1), m, a % 16)
            checked_prefix += 1
        if not ok:
            nb += 1
            if nb <= 3:
                print("MISMATCH round %d stream %d (%d B, loop %d %s): %s" % (r, i, len(streams[i]), loop, plan, why))
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                open(os.path.join(ROOT, "gpurun_out", "prefixfuzz_%d_%d_%d.compressed" % (seed, r, i)), "wb").write(streams[i])
    # nothing behind a capacity: slots are back to back, so a write behind slot i lands in slot i + 1 -- caught above for streams
    # that decode; the tail of the arena for the last one
    if not (arena[int(out_off[-1]):] == 0xEE).all():
        nb += 1
        print("MISMATCH round %d: bytes behind the last slot were written" % r)
    bad += nb
    print("%s round %d: %d streams, loop %d %s" % ("ok" if nb == 0 else "BAD", r, n, loop, plan), flush=True)
print("prefix_fuzz seed %d: %d streams (%d failing ones compared to the oracle's prefix), %d mismatches" % (seed, total, checked_prefix, bad))
sys.exit(1 if bad else 0)
