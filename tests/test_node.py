"""The node in the C ABI (brx_node_*, include/brx.h): one process, one context and one host thread per GPU, a batch dealt
over them.  CPU tests: the dealing (brx_node_deal) against shard.py's, which the gloo tests cover.  GPU tests: 2 .. 4 VIRTUAL ranks
(contexts on device 0: the same code as on four GPUs -- threads, dealing, scatter, gather -- on the one GPU a gpurun box has),
bit-exact against the oracle through host pointers, device pointers with peer copies, and RCCL at world size 1 (the root's shard sent to
itself); the two-GPU test runs wherever two GPUs exist."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

import oracle_py as oracle
from brotli_rs_amd import brx, shard

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))


def _read(name):
    with open(os.path.join(GOLDEN, "data", name), "rb") as f:
        return f.read()


# ---- CPU: the dealing -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,gpus", [(1000, 8), (7, 3), (64, 8), (5, 8), (4097, 4), (1, 1), (0, 4), (65536, 8)])
def test_deal_matches_shard_py(n, gpus):
    rng = np.random.default_rng(n * 31 + gpus)
    sizes = rng.integers(0, 100000, size=n)
    order, cut = brx.node_deal(sizes, gpus, "ranges")
    assert list(order) == list(range(n))
    assert [(int(cut[r]), int(cut[r + 1])) for r in range(gpus)] == shard.shard_ranges(n, gpus)
    order, cut = brx.node_deal(sizes, gpus, "snake")
    assert [(int(cut[r]), int(cut[r + 1])) for r in range(gpus)] == shard.shard_ranges(n, gpus)
    assert list(order) == list(shard.balanced_order(sizes, gpus))
    order, cut = brx.node_deal(sizes, gpus, "bytes")
    assert list(order) == list(range(n)) and cut[0] == 0 and cut[-1] == n and all(cut[r] <= cut[r + 1] for r in range(gpus))


def test_deals_balance_a_ragged_batch():
    """A size-sorted batch (the worst case for index ranges): both balancing deals come within a few percent of equal bytes."""
    rng = np.random.default_rng(5)
    sizes = np.sort(rng.integers(100, 200000, size=4096))[::-1].copy()
    total = int(sizes.sum())
    for deal, tol in (("ranges", None), ("bytes", 1.02), ("snake", 1.08)):
        order, cut = brx.node_deal(sizes, 8, deal)
        shares = [int(sizes[order[cut[r]:cut[r + 1]]].sum()) for r in range(8)]
        assert sum(shares) == total
        if tol is None:
            assert max(shares) > 1.5 * total / 8  # (what the other two are for)
        else:
            assert max(shares) <= tol * total / 8, (deal, shares)
    order, cut = brx.node_deal(sizes, 8, "snake")
    assert sorted(order) == list(range(4096))
    assert all(cut[r + 1] - cut[r] == 512 for r in range(8))  # ... and snake keeps the counts equal as well


def test_deal_rejects_bad_arguments():
    with pytest.raises(brx.BrxError):
        brx.node_deal([1, 2, 3], 0, "ranges")


# ---- GPU ------------------------------------------------------------------------------------------------------------------------------
def _fixture_batch():
    """Every data/ stream (valid ones and reject vectors) + a few repeats: a heterogeneous batch, sizes from 1 B to 400 KB."""
    streams = [_read(e["stream"]) for e in MANIFEST]
    caps = [e["out_bytes"] + 64 if e["status"] == 0 else 1 << 17 for e in MANIFEST]
    return streams, caps, MANIFEST


def _check_fixture_results(outs, status, out_len):
    for e, o, st, ln in zip(MANIFEST, outs, status, out_len):
        assert st == e["status"], (e["stream"], int(st))
        if st == 0:
            assert int(ln) == e["out_bytes"], e["stream"]
            assert hashlib.sha256(o).hexdigest() == e["out_sha256"], e["stream"]


@pytest.mark.gpu
@pytest.mark.parametrize("deal", ["ranges", "bytes", "snake"])
@pytest.mark.parametrize("ranks", [2, 3, 4])
def test_host_pointers_over_virtual_ranks(ranks, deal):
    node = brx.Node([0] * 4)
    try:
        streams, caps, _ = _fixture_batch()
        outs, status, out_len = node.decode_batch(streams, caps, deal=deal, use_gpus=ranks, timing=True)
        _check_fixture_results(outs, status, out_len)
        last = node.last()
        assert last["gpus"] == ranks and sum(last["streams"]) == len(streams) and not last["rccl"]
        assert all(ms > 0 for ms, k in zip(last["kernel_ms"], last["streams"]) if k)
        # a ragged batch in the worst order for index ranges: longest streams first
        rng = random.Random(ranks)
        pool = [_read(n + ".compressed") for n in ("alice29.txt", "asyoulik.txt", "monkey", "quickfox_repeated", "ukkonooa", "x", "lcet10.txt")]
        exp = [_read(n) for n in ("alice29.txt", "asyoulik.txt", "monkey", "quickfox_repeated", "ukkonooa", "x", "lcet10.txt")]
        pick = sorted((rng.randrange(len(pool)) for _ in range(300)), key=lambda i: -len(pool[i]))
        outs, status, out_len = node.decode_batch([pool[i] for i in pick], [len(exp[i]) + 16 for i in pick], deal=deal, use_gpus=ranks)
        assert list(status) == [0] * len(pick)
        assert all(o == exp[i] for o, i in zip(outs, pick))
        shares = node.last()["in_bytes"]
        if deal != "ranges":
            assert max(shares) <= 1.35 * sum(shares) / ranks, shares  # (one lcet10 stream is 4 % of such a batch)
    finally:
        node.close()


@pytest.mark.gpu
def test_pinned_buffers_are_used_in_place_by_every_rank():
    """brx_host_alloc buffers: every rank's kernel reads its slice of the compressed bytes and writes its results in place."""
    node = brx.Node([0, 0])
    try:
        comp, exp = _read("alice29.txt.compressed"), _read("alice29.txt")
        n, cap = 64, (len(exp) + 15) & ~15
        h_in, h_out = brx.host_alloc(n * len(comp)), brx.host_alloc(n * cap)
        h_in[:] = np.frombuffer(comp * n, dtype=np.uint8)
        h_out[:] = 0
        in_off = np.arange(n + 1, dtype=np.uint64) * len(comp)
        out_off = np.arange(n + 1, dtype=np.uint64) * cap
        for deal in ("ranges", "snake"):
            h_out[:] = 0
            status, out_len = node.decode_batch_host_raw(h_in.ctypes.data, in_off, n, h_out.ctypes.data, out_off, deal=deal, use_gpus=2)
            assert list(status) == [0] * n and list(out_len) == [len(exp)] * n
            assert (h_out.reshape(n, cap)[:, :len(exp)] == np.frombuffer(exp, dtype=np.uint8)[None, :]).all()
        brx.host_free(h_in)
        brx.host_free(h_out)
    finally:
        node.close()


def _device_batch(streams, caps):
    import torch
    dev = torch.device("cuda:0")
    n = len(streams)
    in_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(s) for s in streams], out=in_off[1:])
    out_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([(c + 15) & ~15 for c in caps], out=out_off[1:])
    t = {"in": torch.frombuffer(bytearray(b"".join(streams) + b"\0"), dtype=torch.uint8).to(dev),
         "in_off": torch.from_numpy(in_off).to(dev), "out_off": torch.from_numpy(out_off).to(dev),
         "out": torch.zeros(int(out_off[-1]) + 16, dtype=torch.uint8, device=dev),
         "out_len": torch.zeros(n, dtype=torch.int64, device=dev), "status": torch.full((n,), -1, dtype=torch.int32, device=dev)}
    torch.cuda.synchronize()
    return t, out_off


def _run_device(node, t, n, **kw):
    import torch
    t["out"].zero_()
    t["out_len"].zero_()
    t["status"].fill_(-1)
    torch.cuda.synchronize()
    node.decode_batch_device(t["in"].data_ptr(), t["in_off"].data_ptr(), n, t["out"].data_ptr(), t["out_off"].data_ptr(),
                             t["out_len"].data_ptr(), t["status"].data_ptr(), **kw)
    return t["out"].cpu().numpy(), t["status"].cpu().numpy(), t["out_len"].cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("deal", ["ranges", "bytes", "snake"])
def test_device_pointers_scatter_decode_gather_over_virtual_ranks(deal):
    """Everything on the root's GPU; the other ranks get their shard by peer copies, decode into slots of their own, compact, send
    back; the root expands into the caller's slots.  Reject vectors included: status, length and the bytes in front of the error
    are what one context produces on the same batch."""
    node = brx.Node([0] * 4)
    ctx = brx.Context(0)
    try:
        streams, caps, _ = _fixture_batch()
        n = len(streams)
        t, out_off = _device_batch(streams, caps)
        ref_out = ref_status = ref_len = None
        for ranks, root in ((1, 0), (2, 0), (3, 2), (4, 1), (4, 0)):
            out, status, out_len = _run_device(node, t, n, deal=deal, use_gpus=ranks, root=root, timing=(ranks == 3))
            outs = [out[int(out_off[i]):int(out_off[i]) + int(out_len[i])].tobytes() if status[i] == 0 else b"" for i in range(n)]
            _check_fixture_results(outs, status, out_len)
            if ref_out is None:
                ref_out, ref_status, ref_len = out.copy(), status.copy(), out_len.copy()
            else:  # (bytes in front of an error too)
                assert (status == ref_status).all() and (out_len == ref_len).all()
                for i in range(n):
                    k = min(int(out_len[i]), int(out_off[i + 1] - out_off[i]))
                    assert (out[int(out_off[i]):int(out_off[i]) + k] == ref_out[int(out_off[i]):int(out_off[i]) + k]).all(), (ranks, i)
            last = node.last()
            assert last["gpus"] == ranks and sum(last["streams"]) == n and not last["rccl"]
        # the caller's stream is honoured: the batch is produced on it right before the call
        import torch
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            t["in"].copy_(torch.frombuffer(bytearray(b"".join(streams) + b"\0"), dtype=torch.uint8), non_blocking=True)
            out, status, out_len = None, None, None
            node.decode_batch_device(t["in"].data_ptr(), t["in_off"].data_ptr(), n, t["out"].data_ptr(), t["out_off"].data_ptr(),
                                     t["out_len"].data_ptr(), t["status"].data_ptr(), deal=deal, use_gpus=3, hip_stream=s.cuda_stream)
        assert (t["status"].cpu().numpy() == ref_status).all()
    finally:
        ctx.close()
        node.close()


@pytest.mark.gpu
@pytest.mark.parametrize("deal", ["ranges", "snake"])
def test_capacities_are_the_callers_on_every_rank(deal):
    """A slot is its stream's capacity, whatever GPU decodes it: capacities that are no multiple of 16 and a few bytes too small must
    give status 25 and the length needed -- not status 0 and bytes in the neighbour's slot -- exactly as one context does."""
    node = brx.Node([0] * 3)
    ctx = brx.Context(0)
    try:
        names = ("alice29.txt", "asyoulik.txt", "monkey", "ukkonooa", "x", "quickfox_repeated")
        streams = [_read(n + ".compressed") for n in names] * 5
        sizes = [len(_read(n)) for n in names] * 5
        caps = [sz + (7, -1, -5, 3, 0, -13)[i % 6] for i, sz in enumerate(sizes)]
        caps = [max(c, 0) for c in caps]
        ref_outs, ref_status, ref_len = ctx.decode_batch(streams, caps)
        assert 25 in list(ref_status) and 0 in list(ref_status)
        outs, status, out_len = node.decode_batch(streams, caps, deal=deal, use_gpus=3)
        assert list(status) == list(ref_status) and list(out_len) == list(ref_len) and outs == ref_outs
        t, out_off = _device_batch(streams, caps)
        # (exact capacities in the device tables too)
        import torch
        oo = np.zeros(len(caps) + 1, dtype=np.int64)
        np.cumsum(caps, out=oo[1:])
        t["out_off"] = torch.from_numpy(oo).to(t["out"].device)
        t["out"].fill_(0xEE)
        node.decode_batch_device(t["in"].data_ptr(), t["in_off"].data_ptr(), len(streams), t["out"].data_ptr(), t["out_off"].data_ptr(),
                                 t["out_len"].data_ptr(), t["status"].data_ptr(), deal=deal, use_gpus=3, root=1)
        st, ln, out = t["status"].cpu().numpy(), t["out_len"].cpu().numpy(), t["out"].cpu().numpy()
        assert list(st) == list(ref_status) and list(ln) == [int(x) for x in ref_len]
        for i in range(len(streams)):
            if st[i] == 0:
                assert out[int(oo[i]):int(oo[i]) + int(ln[i])].tobytes() == ref_outs[i], i
        assert (out[int(oo[-1]):int(oo[-1]) + 16] == 0xEE).all()  # nothing behind the last slot
    finally:
        ctx.close()
        node.close()


@pytest.mark.gpu
def test_rccl_exchange_at_world_size_one():
    """The RCCL leg -- librccl loaded on demand, ncclCommInitAll, grouped ncclSend / ncclRecv for the scatter and for the ragged
    gather -- on the one GPU of the box: the root's own shard travels through it (BRX_NODE_OPTION_EXCHANGE_ROOT)."""
    node = brx.Node([0], options={"transport": 2, "exchange_root": 1})
    try:
        streams, caps, _ = _fixture_batch()
        n = len(streams)
        t, out_off = _device_batch(streams, caps)
        for deal in ("ranges", "snake"):
            out, status, out_len = _run_device(node, t, n, deal=deal, use_gpus=1)
            outs = [out[int(out_off[i]):int(out_off[i]) + int(out_len[i])].tobytes() if status[i] == 0 else b"" for i in range(n)]
            _check_fixture_results(outs, status, out_len)
            assert node.last()["rccl"]
    finally:
        node.close()
    # RCCL cannot serve ranks that share a GPU: a clear error, not a hang
    node = brx.Node([0, 0], options={"transport": 2})
    try:
        t, out_off = _device_batch(streams[:8], caps[:8])
        with pytest.raises(brx.BrxError):
            _run_device(node, t, 8, use_gpus=2)
    finally:
        node.close()


@pytest.mark.gpu
def test_how_many_gpus_a_batch_is_dealt_over():
    node = brx.Node([0, 0, 0])
    try:
        comp, exp = _read("monkey.compressed"), _read("monkey")
        outs, status, _ = node.decode_batch([comp] * 100, len(exp) + 16)
        assert node.last()["gpus"] == 1 and list(status) == [0] * 100  # fewer streams than one GPU decodes at a time: one GPU
        node.set_option("min_streams", 40)
        outs, status, _ = node.decode_batch([comp] * 100, len(exp) + 16)
        assert node.last()["gpus"] == 3 and node.last()["streams"] == [33, 33, 34] and all(o == exp for o in outs)
        node.set_option("min_streams", 0)
        outs, status, _ = node.decode_batch([comp] * 5000, len(exp) + 16)
        assert node.last()["gpus"] == 2 and all(o == exp for o in outs)  # (16 waves x 256 CUs = 4096 per GPU)
        with pytest.raises(brx.BrxError):
            node.decode_batch([comp] * 4, len(exp) + 16, use_gpus=4)
        # BRX_OPTION_* reach every rank's context
        node.set_option("command_loop", 8)
        outs, status, _ = node.decode_batch([_read("alice29.txt.compressed")] * 9, 160000, use_gpus=3)
        assert all(o == _read("alice29.txt") for o in outs)
    finally:
        node.close()


@pytest.mark.gpu
def test_bad_arguments_are_refused_not_run():
    """Library-level errors (BRX_ERR_INVALID_ARGUMENT with a text in brx_last_error), never a crash: more GPUs than the node has, a root
    outside the ranks used, an unknown deal, NULL tables, decreasing offsets, a device index that does not exist, an unknown option."""
    import ctypes
    L = brx.load_library()
    with pytest.raises(brx.BrxError):
        brx.Node([0, 99])
    node = brx.Node([0, 0])
    try:
        comp, exp = _read("monkey.compressed"), _read("monkey")
        with pytest.raises(brx.BrxError):
            node.decode_batch([comp] * 4, 1024, use_gpus=3)
        t, out_off = _device_batch([comp] * 4, [1024] * 4)
        with pytest.raises(brx.BrxError):
            _run_device(node, t, 4, use_gpus=1, root=1)
        with pytest.raises(brx.BrxError):
            node.set_option("transport", 7)
        with pytest.raises(brx.BrxError):
            node.set_option("command_loop", 3)  # (a BRX_OPTION_* value the contexts refuse)
        opts = brx._NodeOpts(0, 9, 0, 0, None)  # unknown deal
        one = np.zeros(2, dtype=np.uint64)
        assert L.brx_node_decode_batch(node._h, None, one.ctypes.data, 1, None, one.ctypes.data, one.ctypes.data, one.ctypes.data, ctypes.byref(opts)) == -1
        opts = brx._NodeOpts(0, 0, 0, 0, None)
        assert L.brx_node_decode_batch(node._h, None, None, 1, None, None, None, None, ctypes.byref(opts)) == -1  # NULL tables
        assert L.brx_node_decode_batch(node._h, None, None, 0, None, None, None, None, None) == 0  # an empty batch is fine
        bad_off = np.array([5, 3], dtype=np.uint64)
        buf = np.zeros(64, dtype=np.uint8)
        st = np.zeros(1, dtype=np.int32)
        assert L.brx_node_decode_batch(node._h, buf.ctypes.data, bad_off.ctypes.data, 1, buf.ctypes.data, one.ctypes.data, one.ctypes.data, st.ctypes.data,
                                       ctypes.byref(opts)) == -1
        assert b"non-decreasing" in L.brx_last_error()
        assert L.brx_node_ctx(node._h, 5) is None and L.brx_node_size(None) == 0 and L.brx_node_last_timing(None, 0, 0) < 0
        # ... and the node still works afterwards
        outs, status, _ = node.decode_batch([comp] * 4, len(exp) + 16, use_gpus=2)
        assert list(status) == [0] * 4 and all(o == exp for o in outs)
    finally:
        node.close()


@pytest.mark.gpu
def test_two_real_gpus_with_rccl():
    """shard.py's world-2 test in the C ABI: needs two GPUs (the driver's 8-GPU node; skipped on a gpurun box)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    node = brx.Node([0, 1])
    try:
        streams, caps, _ = _fixture_batch()
        n = len(streams)
        outs, status, out_len = node.decode_batch(streams, caps, use_gpus=2)
        _check_fixture_results(outs, status, out_len)
        t, out_off = _device_batch(streams, caps)
        for deal in ("ranges", "snake"):
            for transport in (1, 2):
                node.set_option("transport", transport)
                out, status, out_len = _run_device(node, t, n, deal=deal, use_gpus=2)
                outs = [out[int(out_off[i]):int(out_off[i]) + int(out_len[i])].tobytes() if status[i] == 0 else b"" for i in range(n)]
                _check_fixture_results(outs, status, out_len)
                assert node.last()["rccl"] == (transport == 2)
    finally:
        node.close()


@pytest.mark.gpu
def test_node_from_plain_c(tmp_path):
    """The same through a C++ program that sees nothing but include/brx.h (what a Rust -sys crate binds)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "node_test")
    lib = os.path.join(root, "brotli-rs_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(root, "tests", "cpp", "node_test.cpp"), "-o", exe, "-L", lib, "-lbrx",
                           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, os.path.join(GOLDEN, "data", "alice29.txt.compressed"), os.path.join(GOLDEN, "data", "alice29.txt"), "3", "200"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr
