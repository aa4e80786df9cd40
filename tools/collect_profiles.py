#!/usr/bin/env python3
"""Copy the evidence of tools/gpu_profiles.sh from gpurun_out/ (scratch) into profiles/ (tracked) and rebuild
profiles/hbm_traffic.json, stamped with the id of the kernel sources it measured (bench.py refuses a stale one).
  python tools/collect_profiles.py r02"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
n = 0
for pat in ("bench_%s_*.json", "%s_*_kernel_stats.csv", "%s_sweep.txt", "%s_pcie.txt", "%s_pmc.txt", "%s_spec_ab.txt", "%s_fetchcal.txt", "%s_residency.txt", "%s_plan_ab.txt"):
    for f in glob.glob(os.path.join(G, pat % tag)):
        shutil.copy(f, os.path.join(P, os.path.basename(f)))
        n += 1
import bench  # noqa: E402  (kernel_source_id)
work = {}
for f in glob.glob(os.path.join(G, "traffic_%s_*.json" % tag)):
    t = json.load(open(f))
    fs, ws = t.get("FETCH_SIZE_per_dispatch_raw", []), t.get("WRITE_SIZE_per_dispatch_raw", [])
    if not fs or not ws:
        continue
    fetch, write = sum(fs) / len(fs) * 1024.0, sum(ws) / len(ws) * 1024.0  # counter unit: KiB
    # FETCH_SIZE tallies 64 bytes per 128-byte request of the L2's memory side -- in every access pattern of this kernel
    # (profiles/r03_fetchcal.txt: 1 B per lane scattered over lines, adjacent bytes, dwords, 16 B per lane: ratio 0.500 each;
    # 1 B per lane with every other line untouched: 1.000) -- so the bytes that moved are twice the counter; WRITE_SIZE is exact
    work[t["workload"]] = {"fetch_size_counter_bytes": int(fetch), "fetch_bytes": int(2 * fetch), "write_bytes": int(write),
                           "total_bytes": int(2 * fetch + write)}
json.dump({
    "_note": "HBM-side bytes per launch (brx_decode_kernel + the three wider instances behind it) from rocprofv3 --pmc FETCH_SIZE / "
             "--pmc WRITE_SIZE (separate passes, tools/gpu_profiles.sh), counter unit KiB, mean over the dispatches of one run.  "
             "fetch_bytes = 2 x the FETCH_SIZE counter: the counter tallies 64 bytes per 128-byte memory-side request, measured for "
             "every access pattern of this kernel on known byte counts (tools/ubench/fetchcal.hip, profiles/r03_fetchcal.txt; "
             "MI355X_MICROARCH.md states it for wide streaming reads and asks for this calibration for the others).  "
             "WRITE_SIZE is exact (ratio 1.000).",
    "round": tag, "kernel_source_id": bench.kernel_source_id(), "workloads": work}, open(os.path.join(P, "hbm_traffic.json"), "w"), indent=1)
print("copied %d files; traffic for %s; kernel %s" % (n, sorted(work), bench.kernel_source_id()))
