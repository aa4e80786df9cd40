"""brotli-rs_amd -- MI355X-native batched Brotli decompressor (the decode hot path of ende76/brotli-rs).

The product is the C-ABI shared library `libbrx.so` (include/brx.h) built from csrc/ with hipcc for gfx950.
This Python package is plumbing only: it builds the library (`build_library`) and binds it with ctypes
(`brx`) for the tests, bench.py and __graft_entry__.py.  Import it as `brotli_rs_amd` (repo-root shim).
"""
from .build import build_library, LIB_PATH  # noqa: F401
