"""Shared by the differential fuzzers (wide / big / small / gen): the corrupted variants of a round decoded into slots that are NOT
16-byte aligned (capacity + 0 .. 15 bytes each, from a generator of its own so that the streams of a seed stay what they were), and
compared with the oracle stream by stream: status; length and bytes when the stream decodes; and -- round 6 -- for a stream that
fails, the bytes in FRONT of the error over the shorter of the two lengths (what brx.h now says a slot holds; tools/prefix_fuzz.py
is the fuzz of exactly this over the fixtures, here the encoder's wide / late / big streams get it too)."""
import random

import numpy as np

import oracle_py


def check_corrupted(ctx, cs, cap, seed, report):
    """Returns the number of mismatches; report(i, got_status, want_status, what) is called for each."""
    jit = random.Random(seed * 7919 + len(cs))
    caps = [cap + jit.randrange(16) for _ in cs]
    exp = [oracle_py.decode(s, cap=c) for s, c in zip(cs, caps)]
    n = len(cs)
    in_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(s) for s in cs], out=in_off[1:])
    out_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(caps, out=out_off[1:])
    blob = np.frombuffer(b"".join(cs) + b"\0" * 16, dtype=np.uint8).copy()
    arena = np.zeros(int(out_off[-1]) + 16, dtype=np.uint8)
    status, out_len = ctx.decode_batch_host_raw(blob.ctypes.data, in_off, n, arena.ctypes.data, out_off)
    bad = 0
    for i, (e, st, ln) in enumerate(zip(exp, status, out_len)):
        a, st, ln = int(out_off[i]), int(st), int(ln)
        what = None
        if st != e[0]:
            what = "status"
        elif st == 0:
            if ln != len(e[1]) or arena[a:a + ln].tobytes() != e[1]:
                what = "bytes"
        elif st != 25:
            m = min(ln, len(e[1]), caps[i])
            if arena[a:a + m].tobytes() != e[1][:m]:
                what = "the bytes in front of the error (slot offset %d mod 16)" % (a % 16)
        if what:
            bad += 1
            report(i, st, e[0], what)
    return bad
