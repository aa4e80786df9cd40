#!/usr/bin/env python3
"""Encoder-driven parity fixtures: streams produced by the system libbrotlienc 1.0.9 over a grid of qualities, window
sizes, modes, NPOSTFIX/NDIRECT and forced meta-block flushes.  They reach parts of the format the reference's own
fixtures barely touch (NPOSTFIX/NDIRECT != 0, tens of block types, 1-symbol trees, uncompressed meta-blocks next to
compressed ones).  Each stream is checked here against libbrotlidec AND the oracle before it is written to
tests/golden/enc/ with manifest.json (sha256 of the expected output).
"""
import hashlib
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import brotli_enc  # noqa: E402
import oracle_py  # noqa: E402

G = os.path.join(ROOT, "tests/golden/data")


def sources():
    rng = random.Random(7)
    alice = open(os.path.join(G, "alice29.txt"), "rb").read()
    lcet = open(os.path.join(G, "lcet10.txt"), "rb").read()
    plr = open(os.path.join(G, "plrabn12.txt"), "rb").read()
    src = {}
    src["text40k"] = alice[10000:50000]
    src["text64k"] = lcet[100000:165536]
    src["mixed"] = plr[:12000] + bytes(rng.getrandbits(8) for _ in range(3000)) + b"\x00" * 5000 + plr[5000:17000] + b"abcdefgh" * 900
    src["rand5k"] = bytes(rng.getrandbits(8) for _ in range(5000))
    src["lowent"] = bytes(rng.choice(b"aab") for _ in range(30000))
    src["period"] = (b"0123456789abcdefghijklmnopq" * 3000)[:70000]
    src["binary"] = b"".join(int(i * i * 2654435761 % (1 << 32)).to_bytes(4, "little") + bytes([i & 3, 0, 0, i % 7]) for i in range(6000))
    src["tiny0"] = b""
    src["tiny1"] = b"Z"
    src["tiny10"] = b"hello, hel"
    src["utf8"] = ("größer als — “quoted” naïve café 日本語のテキスト " * 400).encode("utf-8")
    return src


def grid():
    rows = []
    for name in ("text40k", "mixed", "binary", "utf8"):
        for q in (0, 1, 2, 3, 4, 5, 6, 9, 10, 11):
            rows.append((name, dict(quality=q, lgwin=18)))
    for name in ("text64k",):
        for q, lw in ((5, 10), (9, 10), (11, 12), (11, 16), (9, 24), (2, 24), (1, 10)):
            rows.append((name, dict(quality=q, lgwin=lw)))
        rows.append((name, dict(quality=9, lgwin=22, flush_every=8192)))
        rows.append((name, dict(quality=11, lgwin=22, flush_every=20000)))
        rows.append((name, dict(quality=6, lgwin=16, flush_every=3000)))
        for npf, nd in ((1, 2), (2, 12), (3, 40), (0, 15), (3, 120), (1, 30)):
            rows.append((name, dict(quality=9, lgwin=20, npostfix=npf, ndirect=nd)))
            rows.append((name, dict(quality=11, lgwin=18, npostfix=npf, ndirect=nd, mode=brotli_enc.MODE_FONT)))
        rows.append((name, dict(quality=11, lgwin=22, mode=brotli_enc.MODE_TEXT)))
    for name in ("rand5k", "lowent", "period", "tiny0", "tiny1", "tiny10"):
        for q in (0, 2, 5, 9, 11):
            rows.append((name, dict(quality=q, lgwin=16)))
    rows.append(("mixed", dict(quality=10, lgwin=22, flush_every=1000)))
    rows.append(("lowent", dict(quality=11, lgwin=22, flush_every=4096)))
    return rows


def main():
    src = sources()
    d = os.path.join(ROOT, "tests/golden/enc")
    os.makedirs(d, exist_ok=True)
    man = {"generator": "tools/make_enc_fixtures.py (libbrotlienc 1.0.9)", "streams": []}
    total = 0
    for i, (name, kw) in enumerate(grid()):
        data = src[name]
        comp = brotli_enc.compress(data, **kw)
        assert brotli_enc.decompress(comp, len(data)) == data, (name, kw)
        st, out, stats = oracle_py.decode(comp, want_stats=True)
        assert st == 0 and out == data, (name, kw, st)
        fn = "e%03d_%s" % (i, name)
        open(os.path.join(d, fn + ".compressed"), "wb").write(comp)
        total += len(comp)
        man["streams"].append({"name": fn, "params": kw, "in_len": len(comp), "out_len": len(data),
                               "sha256": hashlib.sha256(data).hexdigest(),
                               "meta_blocks": stats.get("meta_blocks"), "commands": stats.get("commands")})
    json.dump(man, open(os.path.join(d, "manifest.json"), "w"), indent=1)
    print(len(man["streams"]), "streams,", total, "compressed bytes")


if __name__ == "__main__":
    main()
