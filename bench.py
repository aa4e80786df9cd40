#!/usr/bin/env python3
"""bench.py -- decompressed MB/s of the batched Brotli decode hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (brx_decode_batch through the C ABI) over one batch of synthetic input
already resident in HBM.  Workload at every N: BASELINE.json configs[1], "4096 x data/alice29.txt.compressed"
PER GPU (weak scaling: independent streams shard across ranks with no data-path collective; the RCCL
scatter/gather of SURVEY 8e lives in brotli-rs_amd/shard.py and is never inside `value`).  Prints ONE JSON line on
rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLD = os.path.join(ROOT, "tests", "golden", "data")
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

C5 = os.path.join(ROOT, "tests", "golden", "config5")
WORKLOADS = {
    # name: (fixtures, streams per GPU); stream i of a batch decodes fixture i mod K
    "alice29x4096": (["alice29.txt"], 4096),                       # BASELINE configs[1]: the headline
    "backward65536x4096": (["backward65536"], 4096),               # configs[2]
    "quickfox_repeatedx8192": (["quickfox_repeated"], 8192),       # configs[3], per-GPU share
    "compressed_repeatedx4096": (["compressed_repeated"], 4096),   # supplementary long-distance copy
    "config5_1MiBx1024": (["c5_0", "c5_1", "c5_2", "c5_3"], 1024), # configs[4], per-GPU share
    # the LZ77 back-reference on its own (north_star: ">= 40 % of HBM peak on the copy"): hand-assembled streams, 64 KiB of
    # raw bytes then non-overlapping copies from distance >= 64 KiB doubling the output to 1 MiB (tests/craft.py)
    "farcopy_1MiBx4096": (["farcopy_0", "farcopy_1", "farcopy_2", "farcopy_3"], 4096),
    # the other Canterbury texts the reference holds (supplementary: bigger files, more trees per meta-block)
    "asyoulikx4096": (["asyoulik.txt"], 4096),
    "lcet10x4096": (["lcet10.txt"], 4096),
    "plrabn12x4096": (["plrabn12.txt"], 4096),
    "mapsdatazrhx4096": (["mapsdatazrh"], 4096),
    # small streams (a launch of many short messages): 47 / 69 / 425 compressed bytes
    "quickfoxx16384": (["quickfox"], 16384),
    "ukkonooax16384": (["ukkonooa"], 16384),
    "monkeyx16384": (["monkey"], 16384),
}
# workloads whose PHYSICAL HBM traffic is known by construction: every copied byte is read from HBM once ("rw") or the
# copy is a periodic fill served from registers / LDS ("w"); for the others the physical figure is the PMC traffic
PHYSICAL_MODEL = {"farcopy_1MiBx4096": "rw", "backward65536x4096": "w", "quickfox_repeatedx8192": "w"}


def load_fixture(name):
    """(compressed, expected).  config-5 fixtures carry only a sha256: their expected bytes come from the oracle."""
    if name.startswith("farcopy_"):
        import craft
        return craft.farcopy_stream(int(name.split("_")[1]))
    if name.startswith("c5_"):
        import hashlib
        import oracle_py
        comp = open(os.path.join(C5, name + ".compressed"), "rb").read()
        st, out = oracle_py.decode(comp)[:2]
        man = {e["name"]: e for e in json.load(open(os.path.join(C5, "manifest.json")))["streams"]}
        assert st == 0 and hashlib.sha256(out).hexdigest() == man[name]["sha256"]
        return comp, out
    return (open(os.path.join(GOLD, name + ".compressed"), "rb").read(), open(os.path.join(GOLD, name), "rb").read())


def cpu_baseline(comp, expect, seconds=12.0):
    """The oracle (CPU restatement of the reference, canonical-lookup mode) on ONE host core, on a bounded
    sample of the same workload: repeated decodes of the same stream for ~`seconds` of CPU time."""
    import ctypes
    import oracle_py
    L = oracle_py.lib()
    cap = len(expect) + 64
    buf = ctypes.create_string_buffer(cap)
    n = ctypes.c_size_t(0)
    st = oracle_py.Stats()
    rc = L.bro_decode(comp, len(comp), buf, cap, ctypes.byref(n), 0, ctypes.byref(st))
    assert rc == 0 and buf.raw[:n.value] == expect
    reps = 0
    t0 = time.perf_counter()
    while True:
        for _ in range(8):
            L.bro_decode(comp, len(comp), buf, cap, ctypes.byref(n), 0, None)
        reps += 8
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    return {"value": round(reps * len(expect) / dt / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "port",
            "sample": "%d sequential decodes of %d B -> %d B in %.1f s, single thread, oracle canonical mode"
                      % (reps, len(comp), len(expect), dt)}, st.as_dict()


def config1_monkey(iters=3000):
    """BASELINE configs[0]: the reference's own bench_monkey plumbing (benches/lib.rs:9-47) on the CPU path that exists
    here, the oracle: the 425-byte stream in memory -> construct the decoder -> read the whole output, ns per iteration
    (median of `iters`), next to the reference's published figure (docs/bench_notes.txt:15, unknown 2015 CPU)."""
    import ctypes
    import oracle_py
    comp, expect = load_fixture("monkey")
    L = oracle_py.lib()
    buf = ctypes.create_string_buffer(len(expect) + 64)
    n = ctypes.c_size_t(0)
    assert L.bro_decode(comp, len(comp), buf, len(buf), ctypes.byref(n), 0, None) == 0 and buf.raw[:n.value] == expect
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter_ns()
        L.bro_decode(comp, len(comp), buf, len(buf), ctypes.byref(n), 0, None)
        ts.append(time.perf_counter_ns() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"workload": "data/monkey.compressed, single stream, CPU path (bench_monkey plumbing)", "ns_per_iter": med,
            "MB_per_s": round(len(expect) / med * 1e3, 1), "iters": iters, "kind": "port",
            "reference_ns_per_iter": 96182, "reference_source": "docs/bench_notes.txt:15 (2015, CPU not stated)",
            "note": "includes the ctypes call overhead (~1 us)"}


def kernel_source_id():
    """sha256 (16 hex digits) of the kernel sources: ties a committed PMC traffic measurement to the kernel it measured."""
    import hashlib
    h = hashlib.sha256()
    for f in ("brx_hot.S", "brx_kernels.hip", "brx_device.h"):
        h.update(open(os.path.join(ROOT, "brotli-rs_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def libbrotlidec_rate(comp, expect, seconds=3.0):
    """Single-thread rate of the system libbrotlidec (1.0.9 in this image) on the same stream -- not the reference, not
    the baseline of record, just a second CPU yardstick next to the oracle.  None when the library is absent."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import ctypes
        import brotli_enc
        if not brotli_enc.available():
            return None
        dec = ctypes.CDLL("libbrotlidec.so.1")
        dec.BrotliDecoderDecompress.argtypes = [ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p]
        out = ctypes.create_string_buffer(len(expect) + 64)
        reps, t0 = 0, time.perf_counter()
        while True:
            for _ in range(8):
                n = ctypes.c_size_t(len(out))
                if dec.BrotliDecoderDecompress(len(comp), comp, ctypes.byref(n), out) != 1:
                    return None
            reps += 8
            dt = time.perf_counter() - t0
            if dt >= seconds:
                break
        if out.raw[:n.value] != expect:
            return None
        return {"value": round(reps * len(expect) / dt / 1e6, 2), "unit": "MB/s", "cores": 1,
                "sample": "%d decodes in %.1f s, libbrotlidec.so.1 one-shot API" % (reps, dt)}
    except OSError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="alice29x4096", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=0, help="override streams per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--verify", type=int, default=1)
    ap.add_argument("--gather", action="store_true", help="also time the ragged gather of the outputs to rank 0 (N>1: always)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from brotli_rs_amd import brx
    ctx = brx.Context(local_rank)

    fixtures, n = WORKLOADS[args.workload]
    if args.streams:
        n = args.streams
    fx = [load_fixture(f) for f in fixtures]
    K = len(fx)
    comp, expect = fx[0]
    cap = (max(len(e) for _, e in fx) + 15) & ~15  # 16-B aligned slots: every stream's flushes are full 16-B stores

    # synthetic batch: stream i = fixture i mod K, each in its own HBM region, distinct output regions
    lens = np.array([len(fx[i % K][0]) for i in range(n)], dtype=np.int64)
    in_off_h = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=in_off_h[1:])
    if K == 1:
        one = torch.frombuffer(bytearray(comp), dtype=torch.uint8).to(dev)
        blob = one.repeat(n).contiguous()
    else:
        parts = [torch.frombuffer(bytearray(c), dtype=torch.uint8).to(dev) for c, _ in fx]
        blob = torch.cat([parts[i % K] for i in range(n)]).contiguous()
    in_off = torch.from_numpy(in_off_h).to(dev)
    out_off = (torch.arange(n + 1, dtype=torch.int64, device=dev) * cap).contiguous()
    out = torch.empty(n * cap, dtype=torch.uint8, device=dev)
    out_len = torch.zeros(n, dtype=torch.int64, device=dev)
    status = torch.full((n,), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def step(timing=False):
        ctx.decode_batch_device(blob.data_ptr(), in_off.data_ptr(), n, out.data_ptr(), out_off.data_ptr(),
                                out_len.data_ptr(), status.data_ptr(), timing=timing)

    def barrier():
        if world > 1:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timing=True)  # HIP events around the kernel, on the stream it is launched on
        kernel_ms.append(ctx.last_timing_ms(1))
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # The RCCL exchange around the decode, timed on its own AFTER the timed region -- never part of `value`:
    # compaction of the capacity slots + grouped send/recv of the ragged outputs to rank 0 (brotli-rs_amd/shard.py).
    gather_info = None
    if world > 1 or args.gather:
        try:
            from brotli_rs_amd import shard
            if world == 1 and not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29533")
                dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
            produced = torch.where(status == 0, out_len, torch.zeros_like(out_len))
            times = []
            for _ in range(3):
                barrier()
                tg = time.perf_counter()
                cdata, coffs = shard.compact(out, out_off, produced)
                full, offs_all, st_all = shard.gather_ragged(cdata, coffs, status, n * world, dst=0, device=dev)
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                times.append(time.perf_counter() - tg)
            g = min(times)
            gather_info = {"ms": round(g * 1e3, 3), "bytes_at_root": int(full.numel()) if full is not None else None,
                           "what": "device-side compaction + grouped send/recv of the ragged outputs to rank 0 (best of 3)"}
            del full
        except Exception as e:  # the gather must never take the decode measurement down with it
            gather_info = {"error": repr(e)[:200]}

    # parity on the timed output: status, lengths, and every stream's bytes (checksum of checksums by equality)
    ok = True
    if args.verify:
        ok = bool((status == 0).all().item())
        for k, (_, e) in enumerate(fx):
            want = torch.frombuffer(bytearray(e), dtype=torch.uint8).to(dev)
            ok = ok and bool((out_len[k::K] == len(e)).all().item())
            got = out.view(n, cap)[k::K, :len(e)]
            ok = ok and bool((got == want.unsqueeze(0)).all().item())
    if world > 1:
        f = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        ok = bool(f.item())

    if rank == 0:
        out_bytes_gpu = sum(len(fx[i % K][1]) for i in range(n))
        total_out = float(out_bytes_gpu) * world
        ms_per_step = dt / args.steps * 1e3
        value = total_out * args.steps / dt / 1e6
        res = {"metric": "decompressed MB/s (whole node), %s batch" % args.workload, "value": round(value, 1),
               "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "u8", "data": "synthetic (fixture(s) %s replicated)" % ",".join(fixtures),
               "config": {"workload": "%d x %s per GPU" % (n, "|".join(f + ".compressed" for f in fixtures)),
                          "streams_per_gpu": n,
                          "in_bytes_per_stream": int(lens.mean()), "out_bytes_per_stream": out_bytes_gpu // n,
                          "sharding": "independent streams, contiguous index range per rank, no data-path collective"},
               "bit_exact": ok}
        if gather_info:
            if "ms" in gather_info:
                gather_info["decode_plus_gather_MB_per_s"] = round(total_out / (dt / args.steps + gather_info["ms"] * 1e-3) / 1e6, 1)
            res["gather"] = gather_info
        import oracle_py
        cb = None
        if not args.no_cpu_baseline:
            cb, _ = cpu_baseline(comp, expect, args.cpu_seconds)
        # ALGORITHMIC bytes per stream: compressed in + decompressed out + window-copy bytes read +
        # dictionary bytes read; per launch = summed over the streams of one GPU.
        algs = []
        for c, e in fx:
            st = oracle_py.decode(c, want_stats=True)[2]
            algs.append(len(c) + len(e) + st["copy_bytes"] + st["dict_bytes"])
        alg_launch = sum(algs[i % K] for i in range(n))
        kms = sorted(kernel_ms)[len(kernel_ms) // 2] if kernel_ms else float("nan")
        kavg = sum(kernel_ms) / max(len(kernel_ms), 1)
        achieved = alg_launch / (kavg * 1e-3) / 1e9
        # HBM-side bytes per launch: PMC counters cannot be read from inside this process; they are collected by
        # tools/gpu_traffic.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, same command) and
        # committed as profiles/hbm_traffic.json.  null when this workload has no committed measurement.
        # A measurement of another kernel version is not reported: traffic = null, traffic_source says "stale".
        traffic, traffic_src = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
            if tj.get("kernel_source_id") == kernel_source_id():
                traffic = tj["workloads"][args.workload]["total_bytes"]
                traffic_src = "profiles/hbm_traffic.json (%s)" % tj.get("round", "?")
            else:
                traffic_src = "stale: profiles/hbm_traffic.json (%s) measured kernel %s, this is %s" % (
                    tj.get("round", "?"), tj.get("kernel_source_id", "?"), kernel_source_id())
        except (OSError, KeyError, ValueError):
            pass
        res["roofline"] = {"bound": "hbm", "kernel": "brx_decode_kernel", "achieved": round(achieved, 1),
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                           "traffic": traffic, "traffic_source": traffic_src,
                           "algorithmic_bytes_per_launch": alg_launch,
                           "kernel_ms_avg": round(kavg, 4), "kernel_ms_median": round(kms, 4)}
        model = PHYSICAL_MODEL.get(args.workload)
        if model:  # physical HBM bytes known by construction ("report both")
            phys = []
            for c, e in fx:
                st = oracle_py.decode(c, want_stats=True)[2]
                phys.append(len(c) + len(e) + (st["copy_bytes"] if model == "rw" else 0))
            phys_launch = sum(phys[i % K] for i in range(n))
            res["roofline"]["physical_bytes_per_launch"] = phys_launch
            res["roofline"]["achieved_physical"] = round(phys_launch / (kavg * 1e-3) / 1e9, 1)
            res["roofline"]["frac_physical"] = round(phys_launch / (kavg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            res["roofline"]["physical_model"] = ("input read + output written + every copied byte read from HBM" if model == "rw"
                                                 else "input read + output written (the fill's source period stays in LDS)")
        elif traffic:
            res["roofline"]["achieved_physical"] = round(traffic / (kavg * 1e-3) / 1e9, 1)
            res["roofline"]["frac_physical"] = round(traffic / (kavg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            res["roofline"]["physical_model"] = "PMC traffic (FETCH_SIZE + WRITE_SIZE)"
        if cb and world == 1 and K == 1:
            # informational, never part of `value`: the same batch from pinned HOST buffers to pinned host buffers through the
            # C ABI's host-pointer path (the kernel reads the input and stores the output over PCIe while it decodes)
            try:
                hin, hout = brx.host_alloc(len(comp) * n), brx.host_alloc(cap * n)
                hin[:] = np.frombuffer(comp * n, dtype=np.uint8)
                io = np.arange(n + 1, dtype=np.uint64) * len(comp)
                oo = np.arange(n + 1, dtype=np.uint64) * cap
                best = 1e9
                for _ in range(3):
                    t1 = time.perf_counter()
                    hst, hln = ctx.decode_batch_host_raw(hin.ctypes.data, io, n, hout.ctypes.data, oo)
                    best = min(best, time.perf_counter() - t1)
                okh = (not hst.any()) and hout[:len(expect)].tobytes() == expect and hout[(n - 1) * cap:(n - 1) * cap + len(expect)].tobytes() == expect
                res["host_path_pcie_inclusive"] = {"ms": round(best * 1e3, 3), "MB_per_s": round(n * len(expect) / best / 1e6, 1),
                                                   "bit_exact": bool(okh), "buffers": "pinned (brx_host_alloc), used in place"}
                brx.host_free(hin)
                brx.host_free(hout)
            except Exception as e:
                res["host_path_pcie_inclusive"] = {"error": repr(e)[:200]}
        if cb:
            res["config1"] = config1_monkey()
            res["cpu_baseline"] = cb
            res["speedup_vs_cpu_1core"] = round(value / world / cb["value"], 1)
            other = libbrotlidec_rate(comp, expect)
            if other:
                res["cpu_libbrotlidec"] = other  # informational: Google's optimized C decoder, if the image has it
            try:  # informational: the oracle on every host core at once (CPU baseline item b)
                import subprocess
                fixture_path = os.path.join(C5 if fixtures[0].startswith("c5_") else GOLD, fixtures[0] + ".compressed")
                o = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_all_cores.py"), fixture_path, "4"],
                                   capture_output=True, text=True, timeout=120)
                res["cpu_all_cores"] = json.loads(o.stdout.strip().splitlines()[-1])
            except Exception:
                pass
    ctx.close()
    if dist.is_initialized():
        dist.destroy_process_group()  # (RCCL prints its library path to stdout around here: the JSON line goes last)
    used_rccl = "gather_info" in dir() and (world > 1 or args.gather)
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(res), flush=True)
    if used_rccl:
        # RCCL prints "Librccl path : ..." to stdout from a destructor at interpreter exit; the contract is ONE JSON line
        # (and it must be the last thing on stdout): leave without running destructors -- everything is flushed and closed.
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0 if ok else 2)
    if not ok:
        sys.exit(2)


if __name__ == "__main__":
    main()
