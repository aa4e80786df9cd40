"""Multi-process (gloo, world_size 2) tests of the N>1 path: stream sharding, ragged scatter of compressed
shards, ragged gather of outputs.  The decode in the middle is the CPU oracle here (tests may use it as the
checker); on a GPU node the same functions run over RCCL with the HIP decode."""
import json
import os
import socket
import sys

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


def _worker(rank, world, port, result_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_py
    from brotli_rs_amd import shard
    manifest = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    names = [e for e in manifest if e["in_bytes"] < 60000]  # keep the CPU suite fast
    streams = [open(os.path.join(GOLDEN, "data", e["stream"]), "rb").read() for e in names] if rank == 0 else None
    caps = [max(e.get("out_bytes", 0), 1 << 17) + 64 for e in names] if rank == 0 else None

    def decode_fn(ss, cs):
        outs, st = [], []
        for s, c in zip(ss, cs):
            rc, o = oracle_py.decode(s, cap=c)
            outs.append(o if rc == 0 else b"")
            st.append(rc)
        return outs, st

    outs, st = shard.decode_sharded(streams, caps, decode_fn, src=0)
    if rank == 0:
        ok = len(outs) == len(names)
        for e, o, s in zip(names, outs, st):
            ok = ok and s == e["status"]
            if e["status"] == 0:
                ok = ok and o == open(os.path.join(GOLDEN, "data", e["expected"]), "rb").read()
        with open(result_path, "w") as f:
            f.write("ok" if ok else "bad")
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


@pytest.mark.timeout(180)
def test_scatter_decode_gather_world2(tmp_path):
    port = _free_port()
    result = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, port, result), nprocs=2, join=True)
    assert open(result).read() == "ok"
