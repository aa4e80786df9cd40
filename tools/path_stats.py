#!/usr/bin/env python3
"""Which command-loop path do the meta-blocks of a set of streams take (BRX_DEBUG_STATS counters)?"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BRX_DEBUG_STATS"] = "1"; os.environ["BRX_DEBUG_STATS_ALL"] = "1"
from brotli_rs_amd import brx
def run(name, streams, caps):
    ctx = brx.Context(0)
    r, w = os.pipe(); saved = os.dup(2); os.dup2(w, 2)
    outs, status, out_len = ctx.decode_batch(streams, caps)
    os.dup2(saved, 2); os.close(w)
    txt = os.read(r, 1 << 24).decode(errors="replace")
    g = f = a = 0
    for m in re.finditer(r"words: (\d+) (\d+) (\d+)", txt):
        g += int(m.group(1)); f += int(m.group(2)); a += int(m.group(3))
    print("%-12s streams %4d  meta-blocks: generic %5d  fast-C++ %5d  asm %5d" % (name, len(streams), g, f, a))
    ctx.close()
G = os.path.join(ROOT, "tests/golden")
man = json.load(open(os.path.join(G, "enc/manifest.json")))["streams"]
run("enc", [open(os.path.join(G, "enc", e["name"] + ".compressed"), "rb").read() for e in man], [e["out_len"] + 16 for e in man])
man = json.load(open(os.path.join(G, "config5/manifest.json")))["streams"]
run("config5", [open(os.path.join(G, "config5", e["name"] + ".compressed"), "rb").read() for e in man], [1 << 20] * len(man))
man = [e for e in json.load(open(os.path.join(G, "manifest.json"))) if e["status"] == 0]
run("reference", [open(os.path.join(G, "data", e["stream"]), "rb").read() for e in man], [e["out_bytes"] + 64 for e in man])
