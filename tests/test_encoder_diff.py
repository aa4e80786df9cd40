"""Encoder-driven differential tests of the oracle (CPU).  Streams come from the system libbrotlienc (tools/
brotli_enc.py): committed ones under tests/golden/enc and tests/golden/config5, plus a seeded random grid generated on
the fly when the library is present.  The oracle must reproduce the original bytes -- on valid streams this pins it
against an independent decoder (libbrotlidec) in addition to the reference's own vectors."""
import hashlib
import json
import os
import random
import sys

import pytest

import oracle_py as oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import brotli_enc  # noqa: E402

ENC = os.path.join(ROOT, "tests", "golden", "enc")
C5 = os.path.join(ROOT, "tests", "golden", "config5")
ENC_MAN = json.load(open(os.path.join(ENC, "manifest.json")))["streams"]
C5_MAN = json.load(open(os.path.join(C5, "manifest.json")))["streams"]


@pytest.mark.parametrize("flags", [0, oracle.FLAG_TREE_WALK], ids=["canonical", "treewalk"])
def test_committed_encoder_streams(flags):
    for e in ENC_MAN:
        comp = open(os.path.join(ENC, e["name"] + ".compressed"), "rb").read()
        st, out = oracle.decode(comp, flags=flags)[:2]
        assert st == 0, e["name"]
        assert len(out) == e["out_len"] and hashlib.sha256(out).hexdigest() == e["sha256"], e["name"]


def test_config5_streams():
    for e in C5_MAN:
        comp = open(os.path.join(C5, e["name"] + ".compressed"), "rb").read()
        st, out, stats = oracle.decode(comp, want_stats=True)
        assert st == 0 and len(out) == 1 << 20 and hashlib.sha256(out).hexdigest() == e["sha256"], e["name"]
        assert stats["meta_blocks"] >= 8  # the forced flushes


@pytest.mark.skipif(not brotli_enc.available(), reason="libbrotlienc/libbrotlidec not in this image")
def test_random_encoder_grid_against_libbrotlidec():
    rng = random.Random(20260928)
    G = os.path.join(ROOT, "tests", "golden", "data")
    pool = [open(os.path.join(G, f), "rb").read() for f in ("alice29.txt", "lcet10.txt", "plrabn12.txt", "asyoulik.txt")]
    for it in range(400):
        kind = rng.randrange(4)
        if kind == 0:
            base = rng.choice(pool)
            o = rng.randrange(len(base) - 40000)
            data = base[o:o + rng.randrange(1, 40000)]
        elif kind == 1:
            data = bytes(rng.getrandbits(8) for _ in range(rng.randrange(0, 3000)))
        elif kind == 2:
            unit = bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 400)))
            data = (unit * (1 + 30000 // len(unit)))[:rng.randrange(1, 30000)]
        else:
            base = rng.choice(pool)
            data = b"".join(base[o:o + 200] for o in (rng.randrange(len(base) - 200) for _ in range(rng.randrange(1, 100))))
        npf = rng.choice([None, 0, 1, 2, 3])
        nd = None if npf is None else rng.randrange(0, 16) << npf
        kw = dict(quality=rng.randrange(0, 12), lgwin=rng.randrange(10, 25), mode=rng.randrange(3), npostfix=npf, ndirect=nd,
                  flush_every=rng.choice([0, 0, 500, 4096, 20000]))
        comp = brotli_enc.compress(data, **kw)
        assert brotli_enc.decompress(comp, len(data)) == data
        st, out = oracle.decode(comp)[:2]
        assert st == 0 and out == data, (it, kw, len(data))
