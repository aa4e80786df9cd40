#!/usr/bin/env python3
"""All-host-cores rate of the CPU oracle on one stream (SURVEY 8d item b): one process per core, each decoding the same
stream for ~`seconds`; prints {"value": MB/s, "cores": n}.  Run as a subprocess by bench.py (no GPU state is inherited)."""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def work(args):
    path, seconds = args
    import ctypes
    import oracle_py
    comp = open(path, "rb").read()
    L = oracle_py.lib()
    cap = 1 << 21
    buf = ctypes.create_string_buffer(cap)
    n = ctypes.c_size_t(0)
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            L.bro_decode(comp, len(comp), buf, cap, ctypes.byref(n), 0, None)
        reps += 4
    return reps * n.value, time.perf_counter() - t0


if __name__ == "__main__":
    path, seconds = sys.argv[1], float(sys.argv[2])
    cores = len(os.sched_getaffinity(0))
    import oracle_py
    oracle_py.lib()  # build once before the workers start
    with mp.Pool(cores) as pool:
        t0 = time.perf_counter()
        res = pool.map(work, [(path, seconds)] * cores)
        wall = time.perf_counter() - t0
    total = sum(r[0] for r in res)
    print(json.dumps({"value": round(total / wall / 1e6, 1), "unit": "MB/s", "cores": cores,
                      "sample": "%d processes x %.1f s, same stream, oracle canonical mode" % (cores, seconds)}))
