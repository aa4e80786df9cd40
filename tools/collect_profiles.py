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
for pat in ("bench_%s_*.json", "%s_*_kernel_stats.csv", "%s_sweep.txt", "%s_pcie.txt", "%s_pmc.txt", "%s_spec_ab.txt"):
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
    work[t["workload"]] = {"fetch_bytes": int(fetch), "write_bytes": int(write), "total_bytes": int(fetch + write),
                           "fetch_bytes_x2": int(2 * fetch)}
json.dump({
    "_note": "HBM-side bytes per brx_decode_kernel launch from rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, "
             "tools/gpu_profiles.sh), counter unit KiB, mean over the dispatches of one run.  FETCH_SIZE is reported uncorrected in "
             "fetch_bytes / total_bytes: the x2 of MI355X_MICROARCH.md applies to wide (16 B/lane) streaming reads -- it does for "
             "farcopy (fetch_bytes_x2 = the 4.0 GB of source bytes + input), not for the 1 B/lane far back-references of text, "
             "where one fetch = one 64-byte line per back-reference.",
    "round": tag, "kernel_source_id": bench.kernel_source_id(), "workloads": work}, open(os.path.join(P, "hbm_traffic.json"), "w"), indent=1)
print("copied %d files; traffic for %s; kernel %s" % (n, sorted(work), bench.kernel_source_id()))
