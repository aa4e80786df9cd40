"""Sharding a batch of independent Brotli streams over the GPUs of one node.

Streams share no state (one reference `Decompressor` owns everything it touches, src/lib.rs:378-394), so the
batch partitions trivially: rank r of G takes the contiguous index range [r*N/G, (r+1)*N/G).  There is NO
collective inside the decode.  RCCL (torch.distributed backend "nccl") is used only to move data in and out:
`scatter_streams` hands every rank its slice of the compressed bytes, `gather_outputs` brings the decoded
streams back to the root.  Both are ragged (sizes differ per rank), hence size exchange + padded tensors.

All functions take a `dist` process group that is already initialised; they work with gloo on CPU tensors
(the multi-process tests) and with nccl on cuda tensors unchanged.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int):
    """Contiguous index range of `rank`: [rank*n//world, (rank+1)*n//world)."""
    return (rank * n) // world, ((rank + 1) * n) // world


def shard_ranges(n: int, world: int):
    return [shard_range(n, r, world) for r in range(world)]


def _dev(device):
    return torch.device(device) if device is not None else torch.device("cpu")


def scatter_streams(streams, src: int = 0, device=None):
    """Root holds `streams` (list of bytes, only read on `src`); every rank receives its shard as a list of
    bytes.  One size exchange (broadcast of the length table) + one scatter of padded byte tensors."""
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = _dev(device)
    if rank == src:
        lens = torch.tensor([len(s) for s in streams], dtype=torch.int64)
        meta = torch.tensor([len(streams)], dtype=torch.int64)
    else:
        meta = torch.zeros(1, dtype=torch.int64)
    meta = meta.to(dev)
    dist.broadcast(meta, src)
    n = int(meta.item())
    lens = lens.to(dev) if rank == src else torch.zeros(n, dtype=torch.int64, device=dev)
    if n:
        dist.broadcast(lens, src)
    lens_h = lens.cpu().numpy()
    ranges = shard_ranges(n, world)
    shard_bytes = [int(lens_h[a:b].sum()) for a, b in ranges]
    pad = max(shard_bytes + [1])
    recv = torch.zeros(pad, dtype=torch.uint8, device=dev)
    if rank == src:
        chunks = []
        for a, b in ranges:
            buf = np.zeros(pad, dtype=np.uint8)
            blob = b"".join(streams[a:b])
            buf[:len(blob)] = np.frombuffer(blob, dtype=np.uint8)
            chunks.append(torch.from_numpy(buf).to(dev))
        dist.scatter(recv, chunks, src=src)
    else:
        dist.scatter(recv, None, src=src)
    a, b = ranges[rank]
    mine = recv.cpu().numpy().tobytes()
    out, at = [], 0
    for ln in lens_h[a:b]:
        out.append(mine[at:at + int(ln)])
        at += int(ln)
    return out, (a, b), n


def gather_outputs(outputs, status, n_total: int, dst: int = 0, device=None):
    """Every rank contributes the decoded streams of its shard (list of bytes) and their status codes; the
    root gets the full lists back in stream order.  Non-root ranks return (None, None)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = _dev(device)
    ranges = shard_ranges(n_total, world)
    a, b = ranges[rank]
    assert len(outputs) == b - a == len(status)
    width = max((hi - lo for lo, hi in ranges), default=0)
    meta = torch.zeros(2 * max(width, 1), dtype=torch.int64, device=dev)
    if b > a:
        meta[:b - a] = torch.tensor([len(o) for o in outputs], dtype=torch.int64)
        meta[width:width + (b - a)] = torch.tensor([int(s) for s in status], dtype=torch.int64)
    metas = [torch.zeros_like(meta) for _ in range(world)] if rank == dst else None
    dist.gather(meta, metas, dst=dst)
    total = torch.tensor([sum(len(o) for o in outputs)], dtype=torch.int64, device=dev)
    dist.all_reduce(total, op=dist.ReduceOp.MAX)
    pad = max(int(total.item()), 1)
    buf = np.zeros(pad, dtype=np.uint8)
    blob = b"".join(outputs)
    buf[:len(blob)] = np.frombuffer(blob, dtype=np.uint8)
    send = torch.from_numpy(buf).to(dev)
    recvs = [torch.zeros(pad, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(send, recvs, dst=dst)
    if rank != dst:
        return None, None
    all_out, all_st = [], []
    for r, (lo, hi) in enumerate(ranges):
        m = metas[r].cpu().numpy()
        raw = recvs[r].cpu().numpy().tobytes()
        at = 0
        for i in range(hi - lo):
            ln = int(m[i])
            all_out.append(raw[at:at + ln])
            all_st.append(int(m[width + i]))
            at += ln
    return all_out, all_st


def decode_sharded(streams, capacities, decode_fn, src: int = 0, device=None):
    """scatter -> local decode -> gather.  `decode_fn(list_of_streams, list_of_capacities)` must return
    (outputs, status); in production it is `brx.Context.decode_batch` (the HIP path)."""
    rank = dist.get_rank()
    shard, (a, b), n = scatter_streams(streams, src=src, device=device)
    caps_t = torch.zeros(max(n, 1), dtype=torch.int64, device=_dev(device))
    if rank == src:
        caps_t[:n] = torch.tensor(list(capacities), dtype=torch.int64)
    dist.broadcast(caps_t, src)
    caps = [int(c) for c in caps_t.cpu().numpy()[a:b]]
    res = decode_fn(shard, caps) if b > a else ([], [])
    outs, st = res[0], res[1]
    return gather_outputs(list(outs), list(st), n, dst=src, device=device)
