#!/usr/bin/env python3
"""Which stream SHAPES are slow per output byte, beyond the committed fixtures (<= 64 KB): pieces of 128 KiB .. 1 MiB of text, an ELF
image, both in turns and low-entropy data through libbrotlienc at quality 1 / 5 / 9 / 11 and lgwin 16 / 22, each as a batch of 256 MiB
of output (4096 x 64 KiB worth of streams), slowest first.  Test tooling (needs the system libbrotlienc).  Usage: gpu_shape_rates.py"""
import glob, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import brx_knobs, brotli_enc
assert brotli_enc.available()
G = os.path.join(ROOT, "tests", "golden", "data")
text = b"".join(open(os.path.join(G, t), "rb").read() for t in ("lcet10.txt", "plrabn12.txt", "alice29.txt", "asyoulik.txt"))
elf = open(sys.executable, "rb").read()
rng = random.Random(3)
lowent = bytes(rng.choice(b"abcdeeeeffgh    \n") for _ in range(1 << 20))
turns = b"".join((text[k:k + 4096] if (k >> 12) & 1 else elf[k:k + 4096]) for k in range(0, 1 << 20, 4096))
dev = torch.device("cuda:0")
ctx = brx_knobs.context(0)
rows = []
for kind, data in (("text", text), ("elf", elf[1 << 16:]), ("turns", turns), ("lowent", lowent)):
    for size in (128 << 10, 512 << 10, 1 << 20):
        for q, lgwin in ((1, 22), (5, 16), (5, 22), (9, 22), (11, 22)):
            piece = data[:size]
            if len(piece) < size:
                continue
            st = brotli_enc.compress(piece, quality=q, lgwin=lgwin)
            n = max(16, (256 << 20) // size)
            cap = (size + 15) & ~15
            blob = torch.frombuffer(bytearray(st), dtype=torch.uint8).to(dev).repeat(n).contiguous()
            in_off = (torch.arange(n + 1, dtype=torch.int64, device=dev) * len(st)).contiguous()
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
            ok = bool((status == 0).all().item()) and out[:size].cpu().numpy().tobytes() == piece
            rows.append((n * size / best / 1e6, kind, size >> 10, q, lgwin, len(st), n, best, ctx.last_wide_streams(1), ctx.last_level4(), ok))
            del blob, out
for r in sorted(rows):
    print("%8.1f GB/s  %-7s %5d KiB q%-2d lgwin %2d  in %7d  x %5d  %9.3f ms  left the regular kernel: %5d, level 4: %4d %s" % (r[:10] + ("" if r[10] else "NOT OK",)), flush=True)
