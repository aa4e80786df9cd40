#!/usr/bin/env python3
"""Write-only yardstick for the periodic fills (configs 3/4): how fast does this GPU take plain stores?  torch fill_ (a
vectorised store kernel) and hipMemsetAsync (rocclr's fill kernel) over buffers of the fills' sizes, next to a device
copy (read + write) of the same size.  One-off, GPU box; numbers quoted in DESIGN.md section 6."""
import time
import torch

dev = torch.device("cuda:0")
for mb in (270, 1440, 4096):
    n = mb * 1000 * 1000
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    v = a.view(torch.int64) if n % 8 == 0 else a
    res = {}
    for name, fn in (("fill_ (int64 view)", lambda: v.fill_(0x0101010101010101)), ("zero_ (memset)", lambda: a.zero_()),
                     ("copy_ (read + write)", lambda: b.copy_(a))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        moved = n * (2 if "copy" in name else 1)
        res[name] = "%.3f ms, %.0f GB/s" % (ms, moved / ms / 1e6)
    print("%d MB:" % mb, res)
