"""ctypes binding of the system libbrotlienc / libbrotlidec (1.0.9 in this image; no headers, no python module).

Test tooling only: it GENERATES streams (tools/make_config5.py, the encoder-driven differential tests) and gives a
second, independent decoder to cross-check the oracle on valid streams.  Nothing in the product path uses it.
"""
import ctypes
import ctypes.util

_enc = _dec = None


def available():
    global _enc, _dec
    if _enc is not None:
        return True
    try:
        _enc = ctypes.CDLL("libbrotlienc.so.1")
        _dec = ctypes.CDLL("libbrotlidec.so.1")
    except OSError:
        _enc = _dec = None
        return False
    c = ctypes
    _enc.BrotliEncoderCreateInstance.restype = c.c_void_p
    _enc.BrotliEncoderCreateInstance.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p]
    _enc.BrotliEncoderSetParameter.argtypes = [c.c_void_p, c.c_int, c.c_uint32]
    _enc.BrotliEncoderCompressStream.argtypes = [c.c_void_p, c.c_int, c.POINTER(c.c_size_t), c.POINTER(c.c_void_p),
                                                 c.POINTER(c.c_size_t), c.POINTER(c.c_void_p), c.POINTER(c.c_size_t)]
    _enc.BrotliEncoderIsFinished.argtypes = [c.c_void_p]
    _enc.BrotliEncoderHasMoreOutput.argtypes = [c.c_void_p]
    _enc.BrotliEncoderDestroyInstance.argtypes = [c.c_void_p]
    _dec.BrotliDecoderDecompress.argtypes = [c.c_size_t, c.c_char_p, c.POINTER(c.c_size_t), c.c_char_p]
    return True


MODE_GENERIC, MODE_TEXT, MODE_FONT = 0, 1, 2
_P_MODE, _P_QUALITY, _P_LGWIN, _P_LGBLOCK = 0, 1, 2, 3
_P_NPOSTFIX, _P_NDIRECT = 7, 8
_OP_PROCESS, _OP_FLUSH, _OP_FINISH = 0, 1, 2


def compress(data, quality=11, lgwin=22, mode=MODE_GENERIC, flush_every=0, npostfix=None, ndirect=None, lgblock=None):
    """One stream; flush_every > 0 forces a meta-block boundary (BROTLI_OPERATION_FLUSH) every that many bytes."""
    assert available()
    c = ctypes
    st = _enc.BrotliEncoderCreateInstance(None, None, None)
    try:
        _enc.BrotliEncoderSetParameter(st, _P_MODE, mode)
        _enc.BrotliEncoderSetParameter(st, _P_QUALITY, quality)
        _enc.BrotliEncoderSetParameter(st, _P_LGWIN, lgwin)
        if lgblock is not None:
            _enc.BrotliEncoderSetParameter(st, _P_LGBLOCK, lgblock)
        if npostfix is not None:
            _enc.BrotliEncoderSetParameter(st, _P_NPOSTFIX, npostfix)
        if ndirect is not None:
            _enc.BrotliEncoderSetParameter(st, _P_NDIRECT, ndirect)
        out = bytearray()
        obuf = c.create_string_buffer(1 << 16)
        src = c.create_string_buffer(bytes(data), len(data)) if len(data) else c.create_string_buffer(1)
        base = c.addressof(src)
        pos = 0
        n = len(data)
        step = flush_every if flush_every > 0 else max(n, 1)
        while True:
            chunk = min(step, n - pos)
            last = pos + chunk >= n
            op = _OP_FINISH if last else (_OP_FLUSH if flush_every > 0 else _OP_PROCESS)
            avail_in = c.c_size_t(chunk)
            next_in = c.c_void_p(base + pos)
            while True:
                avail_out = c.c_size_t(len(obuf))
                next_out = c.c_void_p(c.addressof(obuf))
                ok = _enc.BrotliEncoderCompressStream(st, op, c.byref(avail_in), c.byref(next_in), c.byref(avail_out),
                                                      c.byref(next_out), None)
                if not ok:
                    raise RuntimeError("BrotliEncoderCompressStream failed")
                out += obuf.raw[:len(obuf) - avail_out.value]
                if avail_in.value == 0 and not _enc.BrotliEncoderHasMoreOutput(st):
                    break
            pos += chunk
            if last:
                assert _enc.BrotliEncoderIsFinished(st)
                break
        return bytes(out)
    finally:
        _enc.BrotliEncoderDestroyInstance(st)


def decompress(comp, cap):
    """libbrotlidec one-shot; returns bytes or None on failure."""
    assert available()
    out = ctypes.create_string_buffer(max(cap, 1))
    n = ctypes.c_size_t(cap)
    r = _dec.BrotliDecoderDecompress(len(comp), bytes(comp), ctypes.byref(n), out)
    return out.raw[:n.value] if r == 1 else None
