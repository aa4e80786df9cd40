"""ctypes binding of libbrx.so (include/brx.h) + a `Decompressor` that mirrors the reference's only public
item, `brotli::Decompressor<R: Read>` (reference src/lib.rs:377-410, 2173-2193).

Plumbing only: every decode goes through the C ABI into the gfx950 kernels.  If the library or a GPU is
missing this module FAILS LOUDLY (BrxError); there is no CPU fallback anywhere in the product path.
"""
import ctypes
import io
import os

import numpy as np

from .build import LIB_PATH, build_library

MEM_HOST = 0
MEM_DEVICE = 1
OPT_TIMING = 2
OPT_ORDER = 4
GEN_SWITCHES = 8
GEN_ADAPTIVE = 16

OK = 0
OUTPUT_TOO_SMALL = 25
REF_PANIC = 26


class BrxError(RuntimeError):
    pass


class _Opts(ctypes.Structure):
    _fields_ = [("flags", ctypes.c_uint32), ("reserved", ctypes.c_uint32), ("hip_stream", ctypes.c_void_p)]


class _NodeOpts(ctypes.Structure):  # brx_node_opts
    _fields_ = [("flags", ctypes.c_uint32), ("deal", ctypes.c_uint32), ("use_gpus", ctypes.c_int32), ("root", ctypes.c_int32),
                ("hip_stream", ctypes.c_void_p)]


_lib = None


def load_library():
    """dlopen libbrx.so.  torch is imported first on purpose: its bundled libamdhip64 has the same SONAME
    as the system one, and loading torch first makes both share ONE HIP runtime so device pointers of torch
    tensors are valid inside libbrx."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional for the binding itself
        pass
    path = build_library()
    if not os.path.exists(path):
        raise BrxError("libbrx.so is missing (%s): build it with __graft_entry__.build()" % path)
    L = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    L.brx_ctx_create.restype = ctypes.c_int
    L.brx_ctx_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
    L.brx_ctx_set_option.restype = ctypes.c_int
    L.brx_ctx_set_option.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int64]
    L.brx_ctx_destroy.restype = None
    L.brx_ctx_destroy.argtypes = [ctypes.c_void_p]
    L.brx_decode_batch.restype = ctypes.c_int
    L.brx_decode_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.POINTER(_Opts)]
    L.brx_generate_batch.restype = ctypes.c_int
    L.brx_generate_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                     ctypes.POINTER(_Opts)]
    L.brx_compact_batch.restype = ctypes.c_int
    L.brx_compact_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    L.brx_status_str.restype = ctypes.c_char_p
    L.brx_status_str.argtypes = [ctypes.c_int32]
    L.brx_last_error.restype = ctypes.c_char_p
    L.brx_last_timing.restype = ctypes.c_double
    L.brx_last_trace.restype = ctypes.c_int
    L.brx_last_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    L.brx_last_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.brx_synchronize.restype = ctypes.c_int
    L.brx_synchronize.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.brx_stream_new.restype = ctypes.c_void_p
    L.brx_stream_new.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    L.brx_stream_new_bounded.restype = ctypes.c_void_p
    L.brx_stream_new_bounded.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    L.brx_stream_new_reader.restype = ctypes.c_void_p
    L.brx_stream_new_reader.argtypes = [ctypes.c_void_p, READ_FN, ctypes.c_void_p]
    L.brx_stream_read.restype = ctypes.c_int64
    L.brx_stream_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.brx_stream_free.restype = None
    L.brx_stream_free.argtypes = [ctypes.c_void_p]
    L.brx_host_alloc.restype = ctypes.c_void_p
    L.brx_host_alloc.argtypes = [ctypes.c_size_t]
    L.brx_host_free.restype = None
    L.brx_host_free.argtypes = [ctypes.c_void_p]
    L.brx_node_create.restype = ctypes.c_int
    L.brx_node_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    L.brx_node_destroy.restype = None
    L.brx_node_destroy.argtypes = [ctypes.c_void_p]
    L.brx_node_size.restype = ctypes.c_int
    L.brx_node_size.argtypes = [ctypes.c_void_p]
    L.brx_node_ctx.restype = ctypes.c_void_p
    L.brx_node_ctx.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.brx_node_set_option.restype = ctypes.c_int
    L.brx_node_set_option.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int64]
    L.brx_node_decode_batch.restype = ctypes.c_int
    L.brx_node_decode_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(_NodeOpts)]
    L.brx_node_deal.restype = ctypes.c_int
    L.brx_node_deal.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    L.brx_node_last_timing.restype = ctypes.c_double
    L.brx_node_last_timing.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    _lib = L
    return L


READ_FN = ctypes.CFUNCTYPE(ctypes.c_size_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_ubyte), ctypes.c_size_t)  # brx_read_fn

EXPORTED_SYMBOLS = ["brx_ctx_create", "brx_ctx_destroy", "brx_decode_batch", "brx_status_str", "brx_last_error",
                    "brx_last_timing", "brx_synchronize", "brx_stream_new", "brx_stream_read", "brx_stream_free",
                    "brx_host_alloc", "brx_host_free", "brx_stream_new_bounded", "brx_generate_batch", "brx_compact_batch",
                    "brx_ctx_set_option", "brx_last_trace", "brx_stream_new_reader", "brx_node_create", "brx_node_destroy",
                    "brx_node_size", "brx_node_ctx", "brx_node_set_option", "brx_node_decode_batch", "brx_node_last_timing", "brx_node_deal"]


def status_str(code: int) -> str:
    return load_library().brx_status_str(int(code)).decode("utf-8")


# brx_ctx_set_option (include/brx.h, BRX_OPTION_*): explicit knobs -- neither the library nor this module reads the environment
OPTIONS = {"command_loop": 1, "loop_build": 2, "queue_order": 3, "hand_up": 4, "levels": 5, "tiny_bytes": 6,
           "host_in_place": 7, "grid_cap": 8, "small_bytes": 9, "small_waves": 10, "trace": 11, "reader_window": 12, "level4": 13, "reader_mb_room": 14}


class Context:
    """One brx_ctx: a GPU, its tables, scratch and stream.  `options`: {name: value} of OPTIONS, applied at creation."""

    def __init__(self, device: int = 0, options=None):
        self._lib = load_library()
        h = ctypes.c_void_p()
        rc = self._lib.brx_ctx_create(ctypes.byref(h), device)
        if rc != 0:
            raise BrxError("brx_ctx_create failed (%d): %s" % (rc, self._lib.brx_last_error().decode()))
        self._h = h
        self.device = device
        for name, value in (options or {}).items():
            self.set_option(name, value)

    def set_option(self, name, value):
        rc = self._lib.brx_ctx_set_option(self._h, OPTIONS[name], int(value))
        if rc != 0:
            raise BrxError("brx_ctx_set_option(%s, %d) failed: %s" % (name, value, self._lib.brx_last_error().decode()))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.brx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-memory batch --------------------------------------------------------------------------
    def decode_batch(self, streams, capacities, timing=False):
        """Decode a list of compressed byte strings.  `capacities`: per-stream output capacity (int or list).
        Returns (outputs: list[bytes], status: np.int32[n], out_len: np.uint64[n])."""
        n = len(streams)
        if isinstance(capacities, int):
            capacities = [capacities] * n
        in_off = np.zeros(n + 1, dtype=np.uint64)
        out_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum([len(s) for s in streams], out=in_off[1:])
        np.cumsum(capacities, out=out_off[1:])
        blob = np.frombuffer(b"".join(streams), dtype=np.uint8) if in_off[-1] else np.zeros(1, dtype=np.uint8)
        out = np.zeros(max(int(out_off[-1]), 1), dtype=np.uint8)
        out_len = np.zeros(max(n, 1), dtype=np.uint64)
        status = np.full(max(n, 1), -1, dtype=np.int32)
        opts = _Opts(MEM_HOST | (OPT_TIMING if timing else 0), 0, None)
        rc = self._lib.brx_decode_batch(self._h, blob.ctypes.data, in_off.ctypes.data, n, out.ctypes.data,
                                        out_off.ctypes.data, out_len.ctypes.data, status.ctypes.data,
                                        ctypes.byref(opts))
        if rc != 0:
            raise BrxError("brx_decode_batch failed (%d): %s" % (rc, self._lib.brx_last_error().decode()))
        outs = []
        for i in range(n):
            ln = int(out_len[i]) if status[i] == OK else 0
            outs.append(out[int(out_off[i]):int(out_off[i]) + ln].tobytes())
        return outs, status[:n], out_len[:n]

    # ---- host-memory batch in caller-owned (e.g. pinned) buffers ----------------------------------------
    def decode_batch_host_raw(self, in_ptr, in_off, n, out_ptr, out_off, timing=False):
        """Host pointers as they are (no copies on the Python side): `in_ptr` / `out_ptr` are addresses, e.g. of
        buffers from host_alloc(); in_off / out_off are np.uint64[n+1].  Returns (status, out_len)."""
        out_len = np.zeros(max(n, 1), dtype=np.uint64)
        status = np.full(max(n, 1), -1, dtype=np.int32)
        opts = _Opts(MEM_HOST | (OPT_TIMING if timing else 0), 0, None)
        rc = self._lib.brx_decode_batch(self._h, in_ptr, in_off.ctypes.data, n, out_ptr, out_off.ctypes.data,
                                        out_len.ctypes.data, status.ctypes.data, ctypes.byref(opts))
        if rc != 0:
            raise BrxError("brx_decode_batch failed (%d): %s" % (rc, self._lib.brx_last_error().decode()))
        return status[:n], out_len[:n]

    # ---- stream generator (brx_generate_batch): inputs -> valid Brotli streams, made on the GPU ----------
    @staticmethod
    def generate_slot_bytes(n_bytes, metablock_bytes=65536, adaptive=False):
        """A slot size that always suffices for an input of n_bytes (brx.h)."""
        return n_bytes + n_bytes // 8 + (2048 if adaptive else 256) * (n_bytes // metablock_bytes + 2)

    def generate_batch(self, sources, metablock_bytes=65536, switches=False, adaptive=False):
        """list of bytes -> list of Brotli streams (host buffers; the work happens on the device).  switches: two literal
        block types taking turns every 100 literals (BRX_GEN_SWITCHES).  adaptive: the wave-per-stream generator with codes
        from each meta-block's statistics, two literal trees behind a context map, real block switches (BRX_GEN_ADAPTIVE)."""
        n = len(sources)
        if n == 0:
            return []
        src_off = np.zeros(n + 1, dtype=np.uint64)
        src_off[1:] = np.cumsum([len(x) for x in sources], dtype=np.uint64)
        out_off = np.zeros(n + 1, dtype=np.uint64)
        out_off[1:] = np.cumsum([self.generate_slot_bytes(len(x), metablock_bytes, adaptive) for x in sources], dtype=np.uint64)
        blob = np.frombuffer(b"".join(sources) or b"\0", dtype=np.uint8)
        out = np.zeros(int(out_off[-1]), dtype=np.uint8)
        out_len = np.zeros(n, dtype=np.uint64)
        status = np.full(n, -1, dtype=np.int32)
        opts = _Opts(MEM_HOST | (GEN_SWITCHES if switches else 0) | (GEN_ADAPTIVE if adaptive else 0), 0, None)
        rc = self._lib.brx_generate_batch(self._h, blob.ctypes.data, src_off.ctypes.data, n, out.ctypes.data, out_off.ctypes.data,
                                          out_len.ctypes.data, status.ctypes.data, metablock_bytes, ctypes.byref(opts))
        if rc != 0:
            raise BrxError("brx_generate_batch failed (%d): %s" % (rc, self._lib.brx_last_error().decode()))
        assert not status.any(), status
        return [out[int(out_off[i]):int(out_off[i]) + int(out_len[i])].tobytes() for i in range(n)]

    def generate_batch_device(self, src_ptr, src_off_ptr, n, out_ptr, out_off_ptr, out_len_ptr, status_ptr, metablock_bytes=65536,
                              hip_stream=None, switches=False, adaptive=False):
        """Raw device pointers (e.g. torch tensors): nothing leaves the GPU."""
        opts = _Opts(MEM_DEVICE | (GEN_SWITCHES if switches else 0) | (GEN_ADAPTIVE if adaptive else 0), 0, hip_stream)
        rc = self._lib.brx_generate_batch(self._h, src_ptr, src_off_ptr, n, out_ptr, out_off_ptr, out_len_ptr, status_ptr,
                                          metablock_bytes, ctypes.byref(opts))
        if rc != 0:
            raise BrxError("brx_generate_batch failed (%d): %s" % (rc, self._lib.brx_last_error().decode()))

    # ---- device-memory batch (pointers are raw device addresses, e.g. torch tensor .data_ptr()) ------
    def decode_batch_device(self, in_ptr, in_off_ptr, n, out_ptr, out_off_ptr, out_len_ptr, status_ptr,
                            hip_stream=None, timing=False, order=False):
        opts = _Opts(MEM_DEVICE | (OPT_TIMING if timing else 0) | (OPT_ORDER if order else 0), 0, hip_stream)
        rc = self._lib.brx_decode_batch(self._h, in_ptr, in_off_ptr, n, out_ptr, out_off_ptr, out_len_ptr,
                                        status_ptr, ctypes.byref(opts))
        if rc != 0:
            raise BrxError("brx_decode_batch failed (%d): %s" % (rc, self._lib.brx_last_error().decode()))

    def compact_batch_device(self, out_ptr, out_off_ptr, len_ptr, n, dst_ptr, dst_off_ptr, total, hip_stream=None):
        """dst[dst_off[i] ..] = out[out_off[i] .. + len[i]) for the n streams of a decoded batch (device pointers)."""
        rc = self._lib.brx_compact_batch(self._h, out_ptr, out_off_ptr, len_ptr, n, dst_ptr, dst_off_ptr, total, hip_stream)
        if rc != 0:
            raise BrxError("brx_compact_batch failed (%d): %s" % (rc, self._lib.brx_last_error().decode()))

    def last_timing_ms(self, which=1):
        return float(self._lib.brx_last_timing(self._h, which))

    def last_wide_streams(self, level=1):
        """Streams of the most recent launch decoded at level >= `level` (1..3) of the kernel: their tables spill the LDS table
        memory of the levels below.  Level 1 = every stream that left the regular kernel."""
        return int(self._lib.brx_last_timing(self._h, 1 + level))

    def last_late_streams(self):
        """Streams of the most recent launch handed up at a LATER meta-block with their decoder state (resumed, not restarted)."""
        return int(self._lib.brx_last_timing(self._h, 6))

    def last_redo_bytes(self):
        """Output bytes of the most recent launch that were decoded twice because of hand-overs (0: all of them resumed)."""
        return int(self._lib.brx_last_timing(self._h, 7))

    def last_trace(self, n):
        """(n, 4) uint64: start, end (100 MHz realtime), HW_ID | level << 32, workgroup | grid << 32 per stream (option trace = 1)."""
        import numpy as np
        buf = np.zeros((n, 4), dtype=np.uint64)
        rc = self._lib.brx_last_trace(self._h, buf.ctypes.data, n)
        if rc != 0:
            raise BrxError("brx_last_trace failed (%d): %s" % (rc, self._lib.brx_last_error().decode()))
        return buf

    def stream_short_slices(self):
        """Slices of bounded / pulled streams of this context that paused in front of an item the resident input did not hold."""
        return int(self._lib.brx_last_timing(self._h, 8))

    def last_spec_rollbacks(self):
        """Meta-blocks of the most recent launch taken back after a speculative end (the fast loop read on past the stream's input)."""
        return int(self._lib.brx_last_timing(self._h, 10))

    def last_level4(self):
        """Streams of the most recent launch that level-3 kernels handed on to the level-4 instance (tables beyond 37.6 KiB)."""
        return int(self._lib.brx_last_timing(self._h, 11))

    def slab_waits(self):
        """Waves that ever had to wait for a spill slab on this context (0 by construction since round 6: brx_last_timing 12)."""
        return int(self._lib.brx_last_timing(self._h, 12))

    def pool_slabs(self):
        """Slabs of the context's spill pool right now (brx_last_timing 13)."""
        return int(self._lib.brx_last_timing(self._h, 13))

    def facade_batches(self):
        """(batches, streams) the Read facade has launched for queued streams on this context (brx_last_timing 14 / 15)."""
        return int(self._lib.brx_last_timing(self._h, 14)), int(self._lib.brx_last_timing(self._h, 15))

    def stream_regrown(self):
        """Slices of bounded / pulled streams of this context run again with a larger output buffer (one command beyond the slack)."""
        return int(self._lib.brx_last_timing(self._h, 9))

    def last_lean_listed(self):
        """Streams of the most recent launch that the lean instance (short streams, 32 waves per CU) left to the regular kernel:
        the ones larger than its limit plus the short ones it gave up on (errors, block switches, large tables)."""
        return int(self._lib.brx_last_timing(self._h, 5))

    def synchronize(self, hip_stream=None):
        rc = self._lib.brx_synchronize(self._h, hip_stream)
        if rc != 0:
            raise BrxError("brx_synchronize failed: %s" % self._lib.brx_last_error().decode())

    def decode(self, data: bytes):
        """Decode one stream of unknown size: grow the capacity on OUTPUT_TOO_SMALL.  -> (status, bytes)."""
        cap = max(1 << 16, 8 * len(data))
        while True:
            outs, st, ln = self.decode_batch([data], [cap])
            if st[0] == OUTPUT_TOO_SMALL:
                cap = max(cap * 4, int(ln[0]))
                continue
            return int(st[0]), outs[0]


DEALS = {"ranges": 0, "bytes": 1, "snake": 2}
NODE_OPTIONS = {"transport": 100, "min_streams": 101, "exchange_root": 102}


def node_deal(sizes, gpus, deal="ranges"):
    """brx_node_deal: (order np.uint32[n], cut np.uint32[gpus + 1]) for streams of the given compressed sizes.  No GPU needed."""
    L = load_library()
    n = len(sizes)
    in_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(np.asarray(sizes, dtype=np.uint64), out=in_off[1:])
    order = np.zeros(max(n, 1), dtype=np.uint32)
    cut = np.zeros(gpus + 1, dtype=np.uint32)
    rc = L.brx_node_deal(in_off.ctypes.data, n, gpus, DEALS[deal], order.ctypes.data, cut.ctypes.data)
    if rc != 0:
        raise BrxError("brx_node_deal failed (%d): %s" % (rc, L.brx_last_error().decode()))
    return order[:n], cut


class Node:
    """brx_node: the GPUs of one machine behind one call (one process, one context and one host thread per GPU).  `devices`: HIP
    device index per rank (None = every visible GPU); a device may appear more than once -- "virtual ranks", the same code on one GPU."""

    def __init__(self, devices=None, options=None):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        arr = (ctypes.c_int * len(devices))(*devices) if devices else None
        rc = self._lib.brx_node_create(ctypes.byref(self._h), arr, len(devices) if devices else 0)
        if rc != 0:
            raise BrxError("brx_node_create failed (%d): %s" % (rc, self._lib.brx_last_error().decode()))
        for name, value in (options or {}).items():
            self.set_option(name, value)

    @property
    def size(self):
        return int(self._lib.brx_node_size(self._h))

    def set_option(self, name, value):
        opt = NODE_OPTIONS[name] if name in NODE_OPTIONS else OPTIONS[name]
        rc = self._lib.brx_node_set_option(self._h, opt, int(value))
        if rc != 0:
            raise BrxError("brx_node_set_option(%s) failed (%d): %s" % (name, rc, self._lib.brx_last_error().decode()))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.brx_node_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _opts(self, flags, deal, use_gpus, root, hip_stream, timing):
        return _NodeOpts(flags | (OPT_TIMING if timing else 0), DEALS[deal], int(use_gpus), int(root), hip_stream)

    def decode_batch(self, streams, capacities, deal="ranges", use_gpus=0, timing=False, raw=False):
        """As Context.decode_batch, over the node (host pointers: every GPU reads / writes the caller's buffers itself).  raw=True:
        outputs are the slots' bytes up to min(out_len, capacity) whatever the status (the bytes in front of an error)."""
        n = len(streams)
        if isinstance(capacities, int):
            capacities = [capacities] * n
        in_off = np.zeros(n + 1, dtype=np.uint64)
        out_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum([len(s) for s in streams], out=in_off[1:])
        np.cumsum(capacities, out=out_off[1:])
        blob = np.frombuffer(b"".join(streams), dtype=np.uint8) if in_off[-1] else np.zeros(1, dtype=np.uint8)
        out = np.zeros(max(int(out_off[-1]), 1), dtype=np.uint8)
        out_len = np.zeros(max(n, 1), dtype=np.uint64)
        status = np.full(max(n, 1), -1, dtype=np.int32)
        opts = self._opts(MEM_HOST, deal, use_gpus, 0, None, timing)
        rc = self._lib.brx_node_decode_batch(self._h, blob.ctypes.data, in_off.ctypes.data, n, out.ctypes.data, out_off.ctypes.data,
                                             out_len.ctypes.data, status.ctypes.data, ctypes.byref(opts))
        if rc != 0:
            raise BrxError("brx_node_decode_batch failed (%d): %s" % (rc, self._lib.brx_last_error().decode()))
        outs = []
        for i in range(n):
            ln = min(int(out_len[i]), int(capacities[i])) if (raw or status[i] == OK) else 0
            outs.append(out[int(out_off[i]):int(out_off[i]) + ln].tobytes())
        return outs, status[:n], out_len[:n]

    def decode_batch_host_raw(self, in_ptr, in_off, n, out_ptr, out_off, deal="ranges", use_gpus=0, timing=False):
        """Host pointers as they are (e.g. pinned buffers from host_alloc(): used in place by every GPU).  Returns (status, out_len)."""
        out_len = np.zeros(max(n, 1), dtype=np.uint64)
        status = np.full(max(n, 1), -1, dtype=np.int32)
        opts = self._opts(MEM_HOST, deal, use_gpus, 0, None, timing)
        rc = self._lib.brx_node_decode_batch(self._h, in_ptr, in_off.ctypes.data, n, out_ptr, out_off.ctypes.data,
                                             out_len.ctypes.data, status.ctypes.data, ctypes.byref(opts))
        if rc != 0:
            raise BrxError("brx_node_decode_batch failed (%d): %s" % (rc, self._lib.brx_last_error().decode()))
        return status[:n], out_len[:n]

    def decode_batch_device(self, in_ptr, in_off_ptr, n, out_ptr, out_off_ptr, out_len_ptr, status_ptr, deal="ranges", use_gpus=0,
                            root=0, hip_stream=None, timing=False):
        """Every pointer is memory of the root rank's GPU: scatter over xGMI, decode everywhere, ragged gather.  Returns when done."""
        opts = self._opts(MEM_DEVICE, deal, use_gpus, root, hip_stream, timing)
        rc = self._lib.brx_node_decode_batch(self._h, in_ptr, in_off_ptr, n, out_ptr, out_off_ptr, out_len_ptr, status_ptr,
                                             ctypes.byref(opts))
        if rc != 0:
            raise BrxError("brx_node_decode_batch failed (%d): %s" % (rc, self._lib.brx_last_error().decode()))

    def last(self):
        """Of the most recent batch: {gpus, streams[], in_bytes[], kernel_ms[], wall_ms, scatter_ms, rccl}."""
        t = lambda which, rank=0: float(self._lib.brx_node_last_timing(self._h, which, rank))  # noqa: E731
        g = int(t(0))
        return {"gpus": g, "streams": [int(t(1, r)) for r in range(g)], "in_bytes": [int(t(2, r)) for r in range(g)],
                "kernel_ms": [t(3, r) for r in range(g)], "wall_ms": t(4), "scatter_ms": t(5), "rccl": bool(t(6))}


def host_alloc(nbytes):
    """Pinned host memory from the library (brx_host_alloc) as a numpy uint8 view; free with host_free(view)."""
    L = load_library()
    p = L.brx_host_alloc(nbytes)
    if not p:
        raise BrxError("brx_host_alloc(%d) failed: %s" % (nbytes, L.brx_last_error().decode()))
    arr = np.ctypeslib.as_array((ctypes.c_ubyte * nbytes).from_address(p))
    arr = arr.view(np.uint8)
    _PINNED[arr.ctypes.data] = p
    return arr


def host_free(arr):
    p = _PINNED.pop(arr.ctypes.data, None)
    if p:
        load_library().brx_host_free(p)


_PINNED = {}
_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class Decompressor(io.RawIOBase):
    """Mirror of `brotli::Decompressor<R: Read>` (reference src/lib.rs:377-410): wraps a reader of compressed
    bytes and is itself a reader of decompressed bytes.

        Decompressor(open("x.compressed", "rb")).read()        # == read_to_end

    Semantics of `read` follow the reference's `impl Read` (src/lib.rs:2173-2193): n > 0 bytes, b"" at end of
    stream forever after, and an error carrying the reference's description string for invalid input --
    raised as ValueError here (the reference returns io::ErrorKind::InvalidData).  Differences, documented
    in INTEGRATION.md: the inner reader is drained eagerly on the first read; a stream that fails serves the bytes
    decoded before the error first and raises afterwards (the reference delivers a prefix too).  The object
    keeps its context alive; should the context be closed explicitly first, later reads raise BrxError.
    """

    def __init__(self, reader, ctx: Context = None, streaming: bool = False):
        """streaming=True: the inner reader is PULLED while the stream decodes (brx_stream_new_reader: compressed input and output
        both in bounded sliding windows on the device, any stream length) instead of being drained on the first read."""
        super().__init__()
        self._reader = reader
        self._ctx = ctx
        self._stream = None
        self._lib = load_library()
        self._streaming = streaming
        self._cb = None

    def readable(self):
        return True

    def prepare(self):
        """Drain the inner reader and queue the stream on its context WITHOUT decoding: the first read of any queued
        stream then decodes all of them in one batch (many live Decompressors cost about one batch)."""
        if self._stream is None and self._streaming:
            self._ctx = self._ctx or default_context()
            reader = self._reader

            def pull(_user, buf, cap):
                try:
                    data = reader.read(cap)
                except Exception:  # an exception must not cross the C ABI: the stream ends here (UnexpectedEOF if it was not done)
                    return 0
                n = len(data) if data else 0
                if n:
                    ctypes.memmove(buf, bytes(data), n)
                return n

            self._cb = READ_FN(pull)  # (kept alive as long as the stream)
            self._stream = self._lib.brx_stream_new_reader(self._ctx._h, self._cb, None)
            if not self._stream:
                raise BrxError("brx_stream_new_reader failed")
        if self._stream is None:
            data = self._reader.read() if hasattr(self._reader, "read") else bytes(self._reader)
            self._ctx = self._ctx or default_context()  # (held: the context must outlive the stream object)
            self._stream = self._lib.brx_stream_new(self._ctx._h, data, len(data))
            if not self._stream:
                raise BrxError("brx_stream_new failed")
        return self

    def readinto(self, b):
        self.prepare()
        mv = memoryview(b).cast("B")
        buf = (ctypes.c_ubyte * len(mv)).from_buffer(mv) if len(mv) else None
        n = self._lib.brx_stream_read(self._stream, buf, len(mv))
        if n < -900:
            raise BrxError("libbrx failure: %s" % self._lib.brx_last_error().decode())
        if n < 0:
            raise ValueError(status_str(-n))  # InvalidData + description, src/lib.rs:2177
        return int(n)

    def close(self):
        if self._stream:
            self._lib.brx_stream_free(self._stream)
            self._stream = None
        super().close()
