#!/usr/bin/env python3
"""The many-tree cliff: ONE heterogeneous 4 MiB input (the reference's texts + two ELF images) through libbrotlienc at quality 5 / 9 / 10 /
11, lgwin 24 -> one meta-block with 81 .. 212 literal trees and up to 114 distance trees, ~100 KB of prefix-code tables.  More than 64
trees of a kind, or tables beyond level 3's 37.6 KiB, keep a meta-block out of the assembly loop: it runs in the C++ loop with its tables
in the HBM slab.  Kernel ms of 1 and of 64 copies (device buffers), and the same input cut into 64 KiB streams for comparison.
Usage: gpu_big_trees.py [copies]"""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import brx_knobs, brotli_enc, oracle_py
assert brotli_enc.available()
G = os.path.join(ROOT, "tests", "golden", "data")
src = b"".join(open(f, "rb").read() for f in sorted(glob.glob(os.path.join(G, "*"))) if not f.endswith(".compressed"))
src = (src + open(sys.executable, "rb").read()[:1 << 20] + open(os.path.join(ROOT, "brotli-rs_amd", "libbrx.so"), "rb").read()[:1 << 20])[:4 << 20]
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
ctx = brx_knobs.context(0)


def run(streams, olen, what):
    n = len(streams)
    cap = (olen + 15) & ~15
    blob = torch.frombuffer(bytearray(b"".join(streams)), dtype=torch.uint8).to(dev)
    offs = [0]
    for s in streams:
        offs.append(offs[-1] + len(s))
    in_off = torch.tensor(offs, dtype=torch.int64, device=dev)
    out_off = (torch.arange(n + 1, dtype=torch.int64, device=dev) * cap).contiguous()
    out = torch.empty(n * cap, dtype=torch.uint8, device=dev)
    out_len = torch.zeros(n, dtype=torch.int64, device=dev)
    status = torch.full((n,), -1, dtype=torch.int32, device=dev)
    best = 1e9
    for rep in range(2):
        torch.cuda.synchronize()
        ctx.decode_batch_device(blob.data_ptr(), in_off.data_ptr(), n, out.data_ptr(), out_off.data_ptr(), out_len.data_ptr(), status.data_ptr(), timing=True)
        ctx.synchronize()
        best = min(best, ctx.last_timing_ms(1))
    ok = bool((status == 0).all().item()) and bool((out_len == olen).all().item())
    print("%-58s %5d streams  %10.3f ms  %8.3f GB/s %s" % (what, n, best, n * olen / best / 1e6, "" if ok else "NOT OK"), flush=True)
    return out


text = b"".join(open(os.path.join(G, t), "rb").read() for t in ("lcet10.txt", "plrabn12.txt", "alice29.txt", "asyoulik.txt")) * 3
for name, data, sizes in (("text", text, (1 << 20, 2 << 20)), ("texts + ELF images", src, (4 << 20,))):
    for size in sizes:
        for q in (5, 9, 11):
            piece = data[:size]
            st = brotli_enc.compress(piece, quality=q, lgwin=24)
            r = oracle_py.decode(st, 0, cap=size + 64, want_stats=True)
            assert r[0] == 0 and r[1] == piece
            t0 = time.time(); oracle_py.decode(st, 0, cap=size + 64); cpu = time.time() - t0
            print("%s, %d KiB, quality %d: %d compressed bytes, oracle on one host core %.1f ms; %s" % (name, size >> 10, q, len(st), cpu * 1e3,
                  {k: r[2][k] for k in ("meta_blocks", "commands", "literals", "block_switches") if k in r[2]}), flush=True)
            out = run([st], size, "  one stream")
            assert bytes(out[:size].cpu().numpy().tobytes()) == piece
            run([st] * copies, size, "  %d copies" % copies)
            if os.environ.get("SKIP_CPP") != "1":
                ctx.set_option("command_loop", 6)  # every meta-block as one the assembly loop cannot take: what > 64 trees of a kind meant until round 5
                run([st] * copies, size, "  %d copies, C++ loop (command_loop = 6)" % copies)
                ctx.set_option("command_loop", 0)
    if name != "text":
        parts = [brotli_enc.compress(src[i:i + 65536], quality=9, lgwin=24) for i in range(0, len(src), 65536)]
        run(parts * copies, 65536, "  the same bytes as %d x 64 streams of 64 KiB, quality 9" % copies)
