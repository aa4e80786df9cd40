"""Sharding a batch of independent Brotli streams over the GPUs of one node.

Streams share no state (one reference `Decompressor` owns everything it touches, src/lib.rs:378-394), so the
batch partitions trivially: rank r of G takes the contiguous index range [r*N/G, (r+1)*N/G) -- or, for ragged batches
(`decode_sharded(balance=True)`), the streams dealt by compressed size so that every rank carries about the same bytes.  There is NO
collective inside the decode.  RCCL (torch.distributed backend "nccl") is used only to move data in and out, one
exchange each way, as grouped point-to-point transfers of RAGGED device buffers (`dist.batch_isend_irecv` =
ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd): `scatter_ragged` hands every rank exactly its slice of the
compressed bytes, `gather_ragged` brings the decoded bytes back to the root.  Nothing is padded and nothing takes a
host hop: the buffers are flat uint8 tensors plus int64 offset tables on the device the process group works on.

xGMI is point to point (7 links x ~153 GB/s per GPU): the root's egress/ingress is the bound of either exchange, e.g.
BASELINE config 4 (65536 x quickfox_repeated, 8 GPUs): scatter 0.48 MB per rank, gather 1.44 GB per rank = 10.1 GB
into the root (~9.5 ms over 7 links); config 5 (8192 x 1 MiB): scatter ~360 MB per rank, gather 1.07 GB per rank.

Every function takes the default process group as it is; they run unchanged over gloo with CPU tensors (the
multi-process CPU tests) and over nccl with cuda tensors.  The decode in the middle is any callable with the
device-batch signature of `brx.Context.decode_batch_device` (see `decode_sharded`).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int):
    """Contiguous index range of `rank`: [rank*n//world, (rank+1)*n//world)."""
    return (rank * n) // world, ((rank + 1) * n) // world


def shard_ranges(n: int, world: int):
    return [shard_range(n, r, world) for r in range(world)]


def balanced_order(sizes, world: int):
    """SURVEY 8e for HETEROGENEOUS batches: an order of the streams under which the contiguous ranges of `shard_range` carry about
    equal compressed bytes -- sort by compressed size (largest first: a GPU's time follows its longest streams and its total), deal
    to the ranks in snake order 0 .. G-1, G-1 .. 0, a rank that has its count is skipped.  Returns perm (int64[n]): position k of
    the new order holds stream perm[k] of the caller's.  Homogeneous batches gain nothing from it (the default stays contiguous)."""
    sizes = np.asarray(sizes, dtype=np.int64)
    n = int(sizes.size)
    want = [hi - lo for lo, hi in shard_ranges(n, world)]
    buckets = [[] for _ in range(world)]
    order = np.argsort(-sizes, kind="stable")
    r, step = 0, 1
    for i in order:
        for _ in range(2 * world):  # next rank (snake) that still has room
            if len(buckets[r]) < want[r]:
                break
            nr = r + step
            if nr < 0 or nr >= world:
                step = -step
            else:
                r = nr
        buckets[r].append(int(i))
        nr = r + step
        if nr < 0 or nr >= world:
            step = -step
        else:
            r = nr
    perm = np.array([i for b in buckets for i in b], dtype=np.int64)
    assert perm.size == n and all(len(b) == w for b, w in zip(buckets, want))
    return perm


def reorder_ragged(data, offsets, perm):
    """The ragged batch (data uint8[total], offsets int64[n+1]) with its items in the order perm: (data', offsets').  Index gather on
    the tensors' device, in pieces of <= 64 MiB (as `compact`)."""
    dev = data.device
    perm_t = torch.as_tensor(perm, dtype=torch.int64, device=dev)
    n = int(perm_t.numel())
    lens = (offsets[1:] - offsets[:-1]).to(torch.int64)[perm_t]
    offs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    if n:
        offs[1:] = torch.cumsum(lens, 0)
    total = int(offs[-1].item()) if n else 0
    dst = torch.empty(total, dtype=torch.uint8, device=dev)
    if total == 0:
        return dst, offs
    src_start = offsets[:-1].to(torch.int64)[perm_t]
    offs_h = offs.cpu().numpy()
    piece = 64 << 20
    i = 0
    while i < n:
        j = int(np.searchsorted(offs_h, offs_h[i] + piece, side="right")) - 1
        j = min(max(j, i + 1), n)
        p0, p1 = int(offs_h[i]), int(offs_h[j])
        if p1 > p0:
            idx = torch.repeat_interleave(src_start[i:j] - offs[i:j], lens[i:j]) + torch.arange(p0, p1, dtype=torch.int64, device=dev)
            dst[p0:p1] = data[idx]
        i = j
    return dst, offs


def _dev(device):
    return torch.device(device) if device is not None else torch.device("cpu")


def _exchange(ops):
    """One grouped point-to-point exchange."""
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def scatter_ragged(data, offsets, src: int = 0, device=None):
    """Root holds the concatenated compressed streams `data` (uint8[total]) and `offsets` (int64[n+1]) on `device`
    (both ignored elsewhere).  Every rank gets (data_shard, offsets_shard rebased to 0, (a, b), n): exactly its bytes,
    one broadcast of the offset table + one grouped send/recv."""
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = _dev(device)
    meta = torch.zeros(1, dtype=torch.int64, device=dev)
    if rank == src:
        meta[0] = offsets.numel() - 1
    dist.broadcast(meta, src)
    n = int(meta.item())
    table = offsets.to(dev) if rank == src else torch.zeros(n + 1, dtype=torch.int64, device=dev)
    dist.broadcast(table, src)
    table_h = table.cpu().numpy()
    ranges = shard_ranges(n, world)
    a, b = ranges[rank]
    lo, hi = int(table_h[a]), int(table_h[b])
    ops = []
    if rank == src:
        mine = data[lo:hi]
        for r, (ra, rb) in enumerate(ranges):
            if r != src and table_h[rb] > table_h[ra]:
                ops.append(dist.P2POp(dist.isend, data[int(table_h[ra]):int(table_h[rb])], r))
    else:
        mine = torch.empty(hi - lo, dtype=torch.uint8, device=dev)
        if hi > lo:
            ops.append(dist.P2POp(dist.irecv, mine, src))
    _exchange(ops)
    return mine, (table[a:b + 1] - table[a]).contiguous(), (a, b), n


def compact(out, out_off, out_len, ctx=None):
    """The decoded bytes of a shard without the slack of the capacity slots: (flat uint8 tensor, int64 offsets[n+1]).
    With a `brx.Context` and device tensors: one pass of `brx_compact_batch` (brx_util.hip) at HBM rate, no temporaries.
    Otherwise (CPU tensors of the gloo tests, or no context at hand): a gather by index, done in pieces of <= 64 MiB so the
    int64 index never costs more than ~1.5 GB whatever the shard's size."""
    n = out_len.numel()
    lens = out_len.to(torch.int64)
    offs = torch.zeros(n + 1, dtype=torch.int64, device=out.device)
    if n:
        offs[1:] = torch.cumsum(lens, 0)
    total = int(offs[-1].item()) if n else 0
    if total == 0:
        return torch.empty(0, dtype=torch.uint8, device=out.device), offs
    src_start = out_off[:-1].to(torch.int64).contiguous()
    if ctx is not None and out.is_cuda:
        dst = torch.empty(total, dtype=torch.uint8, device=out.device)
        lens = lens.contiguous()
        torch.cuda.synchronize(out.device)  # the library's stream is not ordered after torch's (brx.h)
        ctx.compact_batch_device(out.data_ptr(), src_start.data_ptr(), lens.data_ptr(), n, dst.data_ptr(), offs.data_ptr(), total)
        return dst, offs
    dst = torch.empty(total, dtype=torch.uint8, device=out.device)
    offs_h = offs.cpu().numpy()
    piece = 64 << 20
    i = 0
    while i < n:
        j = int(np.searchsorted(offs_h, offs_h[i] + piece, side="right")) - 1
        j = min(max(j, i + 1), n)
        p0, p1 = int(offs_h[i]), int(offs_h[j])
        if p1 > p0:
            idx = torch.repeat_interleave(src_start[i:j] - offs[i:j], lens[i:j]) + torch.arange(p0, p1, dtype=torch.int64, device=out.device)
            dst[p0:p1] = out[idx]
        i = j
    return dst, offs


def gather_ragged(data, offsets, status, n_total: int, dst: int = 0, device=None):
    """Every rank contributes the decoded bytes of its shard (`data` uint8[..], `offsets` int64[k+1] rebased to 0) and
    the k status codes.  The root returns (data uint8[total] in stream order, offsets int64[n_total+1], status
    int32[n_total]); the other ranks (None, None, None).  One gather of the small tables + one grouped send/recv."""
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = _dev(device)
    ranges = shard_ranges(n_total, world)
    a, b = ranges[rank]
    k = b - a
    assert offsets.numel() == k + 1 and status.numel() == k
    width = max((hi - lo for lo, hi in ranges), default=0)
    meta = torch.zeros(2 * max(width, 1), dtype=torch.int64, device=dev)
    if k:
        meta[:k] = (offsets[1:] - offsets[:-1]).to(torch.int64)
        meta[width:width + k] = status.to(torch.int64)
    metas = [torch.zeros_like(meta) for _ in range(world)] if rank == dst else None
    dist.gather(meta, metas, dst=dst)
    nbytes = int(offsets[-1].item()) if k else 0
    ops = []
    if rank != dst:
        if nbytes:
            ops.append(dist.P2POp(dist.isend, data[:nbytes].contiguous(), dst))
        _exchange(ops)
        return None, None, None
    lens = torch.cat([metas[r][:hi - lo] for r, (lo, hi) in enumerate(ranges)]) if n_total else torch.zeros(0, dtype=torch.int64, device=dev)
    st = torch.cat([metas[r][width:width + hi - lo] for r, (lo, hi) in enumerate(ranges)]).to(torch.int32) if n_total else torch.zeros(0, dtype=torch.int32, device=dev)
    offs = torch.zeros(n_total + 1, dtype=torch.int64, device=dev)
    if n_total:
        offs[1:] = torch.cumsum(lens, 0)
    offs_h = offs.cpu().numpy()
    full = torch.empty(int(offs_h[-1]), dtype=torch.uint8, device=dev)
    for r, (lo, hi) in enumerate(ranges):
        p0, p1 = int(offs_h[lo]), int(offs_h[hi])
        if p1 == p0:
            continue
        if r == dst:
            full[p0:p1] = data[:nbytes]
        else:
            ops.append(dist.P2POp(dist.irecv, full[p0:p1], r))  # lands in place: contiguous slice of the result
    _exchange(ops)
    return full, offs, st


def decode_sharded(data, offsets, capacities, decode_fn, src: int = 0, device=None, balance: bool = False):
    """scatter -> local decode -> gather, device resident end to end.  balance=True (ragged batches): the root deals the streams to
    the ranks by compressed size (`balanced_order`) instead of by index range, and puts the results back into the caller's order --
    two index gathers on the root, nothing else changes (still no collective inside the decode).

    Root passes `data` uint8[total], `offsets` int64[n+1], `capacities` int64[n] (tensors on `device`).
    `decode_fn(in_t, in_off_t, n, out_t, out_off_t, out_len_t, status_t)` fills the last three for the n streams of the
    shard; on a GPU it is a thin lambda over `brx.Context.decode_batch_device(in_t.data_ptr(), ...)` -- the HIP path --
    and in the CPU tests a stand-in with the same signature.  Returns gather_ragged's result."""
    rank = dist.get_rank()
    dev = _dev(device)
    perm = None
    if balance and rank == src:
        sizes = (offsets[1:] - offsets[:-1]).cpu().numpy()
        perm = balanced_order(sizes, dist.get_world_size())
        data, offsets = reorder_ragged(data, offsets, perm)
        capacities = capacities.to(dev)[torch.as_tensor(perm, dtype=torch.int64, device=dev)]
    shard, offs, (a, b), n = scatter_ragged(data, offsets, src=src, device=dev)
    caps = torch.zeros(max(n, 1), dtype=torch.int64, device=dev)
    if rank == src:
        caps[:n] = capacities.to(dev)
    dist.broadcast(caps, src)
    k = b - a
    my_caps = (caps[a:b] + 15) & ~15  # 16-byte aligned slots: full-width flushes
    out_off = torch.zeros(k + 1, dtype=torch.int64, device=dev)
    if k:
        out_off[1:] = torch.cumsum(my_caps, 0)
    out = torch.empty(int(out_off[-1].item()) if k else 0, dtype=torch.uint8, device=dev)
    out_len = torch.zeros(k, dtype=torch.int64, device=dev)
    status = torch.full((k,), -1, dtype=torch.int32, device=dev)
    if k:
        decode_fn(shard, offs, k, out, out_off, out_len, status)
    produced = torch.where(status == 0, out_len, torch.zeros_like(out_len))  # a failed stream contributes no bytes
    cdata, coffs = compact(out, out_off, produced, ctx=getattr(decode_fn, "ctx", None))
    full, offs_all, st_all = gather_ragged(cdata, coffs, status, n, dst=src, device=dev)
    if perm is not None and full is not None:  # back into the caller's order: position perm[k] takes item k
        inv = np.empty_like(perm)
        inv[perm] = np.arange(perm.size, dtype=np.int64)
        full, offs_all = reorder_ragged(full, offs_all, inv)
        st_all = st_all[torch.as_tensor(inv, dtype=torch.int64, device=st_all.device)]
    return full, offs_all, st_all


# ---- convenience for callers that hold Python byte strings ----------------------------------------------------------
def pack(streams, device=None):
    """list of bytes -> (uint8 tensor, int64 offsets) on `device`."""
    dev = _dev(device)
    lens = np.array([len(s) for s in streams], dtype=np.int64)
    offs = np.zeros(len(streams) + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    blob = np.frombuffer(b"".join(streams), dtype=np.uint8) if offs[-1] else np.zeros(0, dtype=np.uint8)
    return torch.from_numpy(blob.copy()).to(dev), torch.from_numpy(offs).to(dev)


def unpack(data, offsets):
    """(uint8 tensor, int64 offsets) -> list of bytes (host)."""
    raw = data.cpu().numpy().tobytes()
    o = offsets.cpu().numpy()
    return [raw[int(o[i]):int(o[i + 1])] for i in range(len(o) - 1)]


def hip_decode_fn(ctx):
    """The HIP path as a decode_fn for decode_sharded: raw device pointers into brx.Context.decode_batch_device."""
    def fn(in_t, in_off_t, n, out_t, out_off_t, out_len_t, status_t):
        torch.cuda.synchronize(in_t.device)  # the library's stream is not ordered after torch's (brx.h)
        ctx.decode_batch_device(in_t.data_ptr(), in_off_t.data_ptr(), n, out_t.data_ptr(), out_off_t.data_ptr(),
                                out_len_t.data_ptr(), status_t.data_ptr())
        ctx.synchronize()
    fn.ctx = ctx  # (decode_sharded compacts with the library's kernel when the decode is the HIP path)
    return fn
