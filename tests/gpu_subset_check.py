"""Parity subset run in a fresh process (so that BRX_DEBUG_STOP -- read when the context is created -- can select the
C++-only command loops): reference fixtures + encoder fixtures + hand-assembled streams, HIP path vs oracle.
Exit status 0 = all equal.  Used by tests/test_gpu_parity.py::test_cpp_only_command_loops."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import crafted_sets  # noqa: E402
import oracle_py  # noqa: E402
from brotli_rs_amd import brx  # noqa: E402
import brx_knobs  # noqa: E402


def main():
    g = os.path.join(HERE, "golden")
    streams = []
    for e in json.load(open(os.path.join(g, "manifest.json"))):
        streams.append(open(os.path.join(g, "data", e["stream"]), "rb").read())
    for e in json.load(open(os.path.join(g, "enc", "manifest.json")))["streams"]:
        streams.append(open(os.path.join(g, "enc", e["name"] + ".compressed"), "rb").read())
    streams.append(open(os.path.join(g, "config5", "c5_0.compressed"), "rb").read())
    streams += [s for _, s, _, _ in crafted_sets.all_sets()]
    cap = 1 << 20
    want = [oracle_py.decode(s, 0, cap=cap) for s in streams]
    ctx = brx_knobs.context(0)
    outs, status, out_len = ctx.decode_batch(streams, cap)
    bad = [(i, w[0], int(st)) for i, (w, o, st) in enumerate(zip(want, outs, status))
           if w[0] != st or (st == 0 and o != w[1])]
    ctx.close()
    print("checked %d streams, %d mismatches %s" % (len(streams), len(bad), bad[:5]))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
