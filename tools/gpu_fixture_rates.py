#!/usr/bin/env python3
"""Per-fixture rates: every libbrotlienc fixture (tests/golden/enc) x N copies in one device batch -- which stream shapes are slow per
output byte?  Usage: gpu_fixture_rates.py [copies]"""
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import brx_knobs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
man = {e["name"]: e for e in json.load(open(os.path.join(ROOT, "tests", "golden", "enc", "manifest.json")))["streams"]}
dev = torch.device("cuda:0")
ctx = brx_knobs.context(0)
rows = []
for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "enc", "*.compressed"))):
    name = os.path.basename(f)[:-11]
    comp = open(f, "rb").read()
    olen = man[name]["out_len"]
    if olen < 1000:
        continue
    cap = (olen + 15) & ~15
    blob = torch.frombuffer(bytearray(comp), dtype=torch.uint8).to(dev).repeat(n).contiguous()
    in_off = (torch.arange(n + 1, dtype=torch.int64, device=dev) * len(comp)).contiguous()
    out_off = (torch.arange(n + 1, dtype=torch.int64, device=dev) * cap).contiguous()
    out = torch.empty(n * cap, dtype=torch.uint8, device=dev)
    out_len = torch.zeros(n, dtype=torch.int64, device=dev)
    status = torch.full((n,), -1, dtype=torch.int32, device=dev)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        ctx.decode_batch_device(blob.data_ptr(), in_off.data_ptr(), n, out.data_ptr(), out_off.data_ptr(), out_len.data_ptr(), status.data_ptr(), timing=True)
        ctx.synchronize()
        best = min(best, ctx.last_timing_ms(1))
    ok = bool((status == 0).all().item()) and bool((out_len == olen).all().item())
    rows.append((olen * n / best / 1e6, name, man[name]["params"], len(comp), olen, man[name].get("commands"), best, ok))
for r in sorted(rows):
    print("%8.1f GB/s  %-16s %-34s in %7d out %7d cmds %6s  %7.3f ms %s" % (r[0], r[1], json.dumps(r[2]), r[3], r[4], r[5], r[6], "" if r[7] else "NOT OK"))
