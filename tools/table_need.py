#!/usr/bin/env python3
"""(Needs the bring-up build of the library: BRX_BRINGUP=1 python brotli-rs_amd/build.py --force.)
How much table memory do the meta-blocks of large high-quality streams need?  Runs a batch with BRX_NO_DEFER=1 and the
bring-up statistics on and reads each stream's final spill-slab fill (scr_top, words beyond the regular 1 728 words of
LDS table memory, of the stream's LAST meta-block that spilled).  One-off, GPU box."""
import os
import random
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["BRX_NO_DEFER"] = "1"
os.environ["BRX_DEBUG_STATS"] = "1"
os.environ["BRX_DEBUG_STATS_ALL"] = "1"
import brotli_enc  # noqa: E402
from brotli_rs_amd import brx  # noqa: E402
import brx_knobs  # noqa: E402

G = os.path.join(ROOT, "tests", "golden", "data")
rng = random.Random(3)
corpus = b"".join(open(os.path.join(G, f), "rb").read() for f in ("lcet10.txt", "plrabn12.txt", "alice29.txt", "asyoulik.txt", "mapsdatazrh"))
rows = []
streams, caps, tags = [], [], []
for size in (100000, 200000, 400000, 800000, len(corpus)):
    for q in (5, 9, 10, 11):
        for rep in range(3):
            o = rng.randrange(len(corpus) - size + 1)
            d = corpus[o:o + size]
            streams.append(brotli_enc.compress(d, quality=q, lgwin=24))
            caps.append(len(d) + 16)
            tags.append((size, q))
for name in ("lcet10.txt", "plrabn12.txt", "alice29.txt", "asyoulik.txt", "mapsdatazrh"):
    streams.append(open(os.path.join(G, name + ".compressed"), "rb").read())
    caps.append(len(open(os.path.join(G, name), "rb").read()) + 16)
    tags.append((name, "fixture"))
ctx = brx_knobs.context(0)
r, w = os.pipe(); saved = os.dup(2); os.dup2(w, 2)
outs, status, out_len = ctx.decode_batch(streams, caps)
os.dup2(saved, 2); os.close(w)
txt = b""
while True:
    b = os.read(r, 1 << 20)
    txt += b
    if len(b) < (1 << 20):
        break
top = [int(m.group(1)) for m in re.finditer(r"scr_top=(\d+)", txt.decode(errors="replace"))]
assert len(top) == len(streams), (len(top), len(streams))
for t, s_, st in zip(tags, top, status):
    print(t, "status", int(st), "words beyond 1728:", s_, "->", "fits 10 KiB" if s_ == 0 else "fits 13 KiB" if s_ <= 768 else
          "fits 20 KiB" if s_ <= 2560 else "fits 40 KiB" if s_ <= 7680 else "beyond 40 KiB")
