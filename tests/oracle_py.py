"""ctypes binding of the CPU oracle (oracle/brotli_oracle.c).  Test infrastructure only."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_build", "libbrotli_oracle.so")

FLAG_TREE_WALK = 1
STATUS_OUTPUT_TOO_SMALL = 25


class Stats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in (
        "meta_blocks", "commands", "literals", "raw_bytes", "copies", "copy_bytes", "overlapped_copies",
        "dict_refs", "dict_bytes", "block_switches", "bits_consumed", "max_distance")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_lib = None


def build():
    """Compile the oracle if the shared object is missing or older than its sources."""
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("brotli_oracle.c", "brotli_oracle.h", "Makefile")]
    if (not os.path.exists(LIB)) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB)
        L.bro_decode.restype = ctypes.c_int
        L.bro_decode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                 ctypes.POINTER(ctypes.c_size_t), ctypes.c_uint, ctypes.POINTER(Stats)]
        L.bro_status_str.restype = ctypes.c_char_p
        L.bro_status_str.argtypes = [ctypes.c_int]
        L.bro_transform.restype = ctypes.c_int
        L.bro_transform.argtypes = [ctypes.c_uint, ctypes.c_char_p, ctypes.c_uint, ctypes.c_void_p]
        L.bro_inverse_mtf.restype = None
        L.bro_inverse_mtf.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.bro_dictionary.restype = ctypes.POINTER(ctypes.c_ubyte)
        L.bro_context_lut.restype = ctypes.POINTER(ctypes.c_ubyte)
        L.bro_context_lut.argtypes = [ctypes.c_int]
        L.bro_insert_copy_entry.restype = None
        L.bro_insert_copy_entry.argtypes = [ctypes.c_uint] + [ctypes.POINTER(ctypes.c_uint32)] * 4
        _lib = L
    return _lib


def decode(data: bytes, flags: int = 0, cap: int = None, want_stats: bool = False):
    """Decode one stream.  Returns (status, output_bytes[, stats]).  Grows the buffer on status 25
    unless an explicit `cap` is given."""
    L = lib()
    fixed = cap is not None
    cap = cap if fixed else max(1 << 16, 8 * len(data))
    while True:
        buf = ctypes.create_string_buffer(max(cap, 1))
        n = ctypes.c_size_t(0)
        st = Stats()
        rc = L.bro_decode(data, len(data), buf, cap, ctypes.byref(n), flags, ctypes.byref(st))
        if rc == STATUS_OUTPUT_TOO_SMALL and not fixed:
            cap = max(cap * 4, n.value)
            continue
        out = buf.raw[:min(n.value, cap)]
        return (rc, out, st.as_dict()) if want_stats else (rc, out)


def status_str(code: int) -> str:
    return lib().bro_status_str(code).decode("utf-8")


def transform(tid: int, word: bytes):
    buf = ctypes.create_string_buffer(64)
    n = lib().bro_transform(tid, word, len(word), buf)
    return None if n < 0 else buf.raw[:n]


def inverse_mtf(v: bytes) -> bytes:
    buf = ctypes.create_string_buffer(bytes(v), len(v))
    lib().bro_inverse_mtf(buf, len(v))
    return buf.raw[:len(v)]
