#!/usr/bin/env python3
"""(Needs the bring-up build of the library: BRX_BRINGUP=1 python brotli-rs_amd/build.py --force.)
Which command-loop path do the meta-blocks of a set of streams take (BRX_DEBUG_STATS counters)?"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["BRX_DEBUG_STATS"] = "1"; os.environ["BRX_DEBUG_STATS_ALL"] = "1"
from brotli_rs_amd import brx
def run(name, streams, caps):
    ctx = brx.Context(0)
    r, w = os.pipe(); saved = os.dup(2); os.dup2(w, 2)
    outs, status, out_len = ctx.decode_batch(streams, caps)
    os.dup2(saved, 2); os.close(w)
    txt = os.read(r, 1 << 24).decode(errors="replace")
    g = a = x0 = x1 = x2 = 0
    import collections
    why = collections.Counter()
    for m in re.finditer(r"words: (\d+) (\d+) (\d+) (\d+) (\d+) (\d+) (\d+) (\d+) (\d+)", txt):
        g += int(m.group(1)); a += int(m.group(3)); x0 += int(m.group(5)); x1 += int(m.group(6)); x2 += int(m.group(7))
        w = int(m.group(9))
        for bit, nm in ((1, "handle table beyond LDS"), (2, "a tree beyond LDS"), (4, "empty / one-symbol insert&copy tree"), (8, "incomplete code"), (16, "> 64 trees"), (32, "capacity / maps beyond LDS")):
            if w & bit:
                why[nm] += 1
    import oracle_py
    cmds = sum(oracle_py.decode(s_, want_stats=True)[2]["commands"] for s_ in streams)
    print("%-10s streams %4d  meta-blocks: C++ only %4d, assembly %5d;  commands %8d, exits to C++ at R0 %6d  R1 %6d  R2 %6d  (%.2f %% of commands)"
          % (name, len(streams), g, a, cmds, x0, x1, x2, 100.0 * (x0 + x1 + x2) / max(cmds, 1)))
    if why:
        print("           streams with a C++-only meta-block, by reason:", dict(why))
    ctx.close()
G = os.path.join(ROOT, "tests/golden")
man = json.load(open(os.path.join(G, "enc/manifest.json")))["streams"]
run("enc", [open(os.path.join(G, "enc", e["name"] + ".compressed"), "rb").read() for e in man], [e["out_len"] + 16 for e in man])
man = json.load(open(os.path.join(G, "config5/manifest.json")))["streams"]
run("config5", [open(os.path.join(G, "config5", e["name"] + ".compressed"), "rb").read() for e in man], [1 << 20] * len(man))
man = [e for e in json.load(open(os.path.join(G, "manifest.json"))) if e["status"] == 0]
run("reference", [open(os.path.join(G, "data", e["stream"]), "rb").read() for e in man], [e["out_bytes"] + 64 for e in man])
