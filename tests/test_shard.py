"""Multi-process tests of the N>1 path: stream sharding, grouped point-to-point scatter of the ragged
compressed shards, local decode through the device-batch signature, ragged gather of the outputs.

CPU (not gpu): world_size 2 over gloo with CPU tensors; the decode in the middle is a stand-in with exactly the
signature of the HIP path (flat tensors + offset tables, results written in place) that calls the oracle -- tests may
use it as the checker.  GPU: the same functions with `shard.hip_decode_fn` (the HIP path): world 1 on any box, world 2
over RCCL when the box has two GPUs."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fixtures(limit=60000):
    manifest = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    return [e for e in manifest if e["in_bytes"] < limit]  # keep the CPU suite fast


def _oracle_decode_fn():
    """Stand-in for the HIP path with its signature: decodes the shard's streams with the oracle into the slots."""
    import oracle_py

    def fn(in_t, in_off_t, n, out_t, out_off_t, out_len_t, status_t):
        raw = in_t.cpu().numpy().tobytes()
        io, oo = in_off_t.cpu().numpy(), out_off_t.cpu().numpy()
        for i in range(n):
            cap = int(oo[i + 1] - oo[i])
            rc, o = oracle_py.decode(raw[int(io[i]):int(io[i + 1])], cap=cap)
            status_t[i] = rc
            out_len_t[i] = len(o)
            if len(o):
                out_t[int(oo[i]):int(oo[i]) + len(o)] = torch.from_numpy(np.frombuffer(o, dtype=np.uint8).copy())
    return fn


def _check(names, data, offs, st):
    from brotli_rs_amd import shard
    outs = shard.unpack(data, offs)
    ok = len(outs) == len(names) and st.numel() == len(names)
    for e, o, s in zip(names, outs, st.cpu().tolist()):
        ok = ok and s == e["status"]
        if e["status"] == 0:
            ok = ok and o == open(os.path.join(GOLDEN, "data", e["expected"]), "rb").read()
        else:
            ok = ok and o == b""
    return ok


def _worker(rank, world, port, result_path, backend, balance=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dev = torch.device("cpu")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from brotli_rs_amd import shard
    names = _fixtures()
    if rank == 0:
        data, offs = shard.pack([open(os.path.join(GOLDEN, "data", e["stream"]), "rb").read() for e in names], dev)
        caps = torch.tensor([max(e.get("out_bytes", 0), 1 << 17) + 64 for e in names], dtype=torch.int64, device=dev)
    else:
        data = offs = caps = None
    if backend == "nccl":
        from brotli_rs_amd import brx
        ctx = brx.Context(rank)
        fn = shard.hip_decode_fn(ctx)
    else:
        fn = _oracle_decode_fn()
    out, out_offs, st = shard.decode_sharded(data, offs, caps, fn, src=0, device=dev, balance=balance)
    if rank == 0:
        with open(result_path, "w") as f:
            f.write("ok" if _check(names, out, out_offs, st) else "bad")
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_partition_exactly():
    from brotli_rs_amd import shard
    for n in (0, 1, 7, 8, 4096, 65536, 65537):
        for world in (1, 2, 3, 4, 8):
            r = shard.shard_ranges(n, world)
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_compact_drops_the_slack_of_the_slots():
    from brotli_rs_amd import shard
    out = torch.arange(100, dtype=torch.uint8)
    out_off = torch.tensor([0, 10, 40, 40, 100], dtype=torch.int64)
    out_len = torch.tensor([3, 0, 0, 7], dtype=torch.int64)
    data, offs = shard.compact(out, out_off, out_len)
    assert offs.tolist() == [0, 3, 3, 3, 10]
    assert data.tolist() == [0, 1, 2, 40, 41, 42, 43, 44, 45, 46]


@pytest.mark.timeout(180)
def test_scatter_decode_gather_world2(tmp_path):
    port = _free_port()
    result = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, port, result, "gloo"), nprocs=2, join=True)
    assert open(result).read() == "ok"


@pytest.mark.timeout(180)
def test_scatter_decode_gather_world3_uneven(tmp_path):
    """43 + 9 streams over 3 ranks: uneven shards, an empty exchange never hangs."""
    port = _free_port()
    result = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(3, port, result, "gloo"), nprocs=3, join=True)
    assert open(result).read() == "ok"


def test_balanced_order_deals_by_compressed_size():
    """ragged batches are dealt by size.  A batch sorted by size (the worst case for contiguous ranges) ends up within a
    few percent of equal bytes per rank; the counts are those of the contiguous ranges; every stream appears exactly once."""
    from brotli_rs_amd import shard
    rng = np.random.default_rng(5)
    for n, world in ((52, 2), (52, 3), (4096, 8), (7, 8), (0, 4), (1000, 3)):
        sizes = np.sort(rng.integers(10, 500000, size=n))
        perm = shard.balanced_order(sizes, world)
        assert sorted(perm.tolist()) == list(range(n))
        loads, contiguous = [], []
        for lo, hi in shard.shard_ranges(n, world):
            loads.append(int(sizes[perm[lo:hi]].sum()))
            contiguous.append(int(sizes[lo:hi].sum()))
        if n >= 50:
            assert max(loads) <= 1.08 * (sum(loads) / world), (n, world, loads)
            assert max(contiguous) > 1.3 * (sum(loads) / world)  # (what the index ranges would have given)


def test_reorder_ragged_round_trip():
    from brotli_rs_amd import shard
    items = [bytes([i]) * (i * 7 % 23) for i in range(40)]
    data, offs = shard.pack(items)
    perm = np.random.default_rng(1).permutation(len(items))
    d2, o2 = shard.reorder_ragged(data, offs, perm)
    assert shard.unpack(d2, o2) == [items[i] for i in perm]
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(items))
    d3, o3 = shard.reorder_ragged(d2, o2, inv)
    assert shard.unpack(d3, o3) == items


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 3])
def test_scatter_decode_gather_balanced(tmp_path, world):
    """The same ragged fixture batch dealt by compressed size: results come back in the caller's order."""
    port = _free_port()
    result = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, port, result, "gloo", True), nprocs=world, join=True)
    assert open(result).read() == "ok"


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_sharded_decode_hip_path_world1(tmp_path):
    """The exact code path of a multi-GPU run (device-resident ragged tensors, the HIP decode through raw device
    pointers, compaction and gather on the device) with a process group of one rank over RCCL."""
    port = _free_port()
    result = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(1, port, result, "nccl"), nprocs=1, join=True)
    assert open(result).read() == "ok"


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_sharded_decode_hip_path_world2_rccl(tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU tier)")
    port = _free_port()
    result = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, port, result, "nccl"), nprocs=2, join=True)
    assert open(result).read() == "ok"
