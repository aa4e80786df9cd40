"""CPU-side checks of the drop-in boundary: libbrx.so builds for gfx950, loads, exports every symbol that
include/brx.h declares, and refuses to run without a GPU (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

import brotli_rs_amd
from brotli_rs_amd import brx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    hdr = open(os.path.join(ROOT, "include", "brx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(brx_[a-z_]+)\s*\(", hdr)))


def test_library_builds_for_gfx950():
    path = brotli_rs_amd.build_library()
    assert os.path.exists(path)
    blob = open(path, "rb").read()
    assert b"gfx950" in blob  # the code object targets MI355X
    assert b"brx_decode_kernel" in blob
    # the wider-LDS instances of the kernel (brx_device.h, BRX_LEVEL) travel in the same library
    for k in (1, 2, 3, 4):
        assert b"brx_decode_kernel_l%d" % k in blob
    assert b"brx_decode_kernel_s" in blob  # ... and the lean instance for short streams (brx_small.h)


def test_shipped_library_reads_no_environment():
    """the reference has no environment knobs; here the C ABI takes explicit arguments
    (brx_ctx_set_option).  The default build of libbrx.so must not name a BRX_* environment variable nor import getenv
    (the bring-up build, BRX_BRINGUP=1, may)."""
    import subprocess
    path = brotli_rs_amd.build_library()
    blob = open(path, "rb").read()
    former = [b"BRX_DEBUG_STOP", b"BRX_DEBUG_STATS", b"BRX_DEBUG_DUMP", b"BRX_NO_ORDER", b"BRX_NO_DEFER", b"BRX_GRID_CAP", b"BRX_NO_OVERLAP",
              b"BRX_FORCE_OVERLAP", b"BRX_PLAN_A", b"BRX_PLAN_B", b"BRX_TINY_BYTES", b"BRX_NO_MIRROR", b"BRX_LOOP_BUILD", b"BRX_SMALL_BYTES", b"BRX_SMALL_WAVES"]
    assert not [n for n in former if n in blob]
    nm = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True)
    assert nm.returncode == 0 and "getenv" not in nm.stdout


def test_every_declared_symbol_is_exported():
    lib = brx.load_library()
    declared = _declared_functions()
    assert set(declared) == set(brx.EXPORTED_SYMBOLS), declared
    for name in declared:
        assert getattr(lib, name) is not None


def test_status_strings_are_the_reference_descriptions():
    """1..24 = DecompressorError descriptions, reference src/lib.rs:331-354 (typos included)."""
    import oracle_py
    for code in range(0, 28):
        assert brx.status_str(code) == oracle_py.status_str(code)
    assert brx.status_str(24) == "Encountered unexpected EOF"
    assert brx.status_str(15) == "Enocuntered non-zero bit trailing the stream"


def test_no_cpu_fallback_without_gpu():
    """On a box without a HIP device the product path must fail loudly, never decode on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(brx.BrxError):
        brx.Context(0)
    with pytest.raises(brx.BrxError):
        brx.Decompressor(open(os.path.join(ROOT, "tests", "golden", "data", "64x.compressed"), "rb")).read()


def test_file_walker_is_built_and_has_no_cpu_path():
    """brx_walk (the reference's file walker on the batched decoder) is built with the library; without a GPU it stops at
    brx_ctx_create and says so."""
    import subprocess
    import torch
    exe = os.path.join(ROOT, "brotli-rs_amd", "brx_walk")
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "data")], capture_output=True, text=True)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr, r.stderr


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under brotli-rs_amd/ may import, include or link it."""
    pkg = os.path.join(ROOT, "brotli-rs_amd")
    for dirpath, _, files in os.walk(pkg):
        if "_gen" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "brotli_oracle" not in text and "oracle_py" not in text, os.path.join(dirpath, f)
                assert "libbrotli" not in text, os.path.join(dirpath, f)


def test_assembly_loop_passes_the_wait_state_lint():
    """brx_hot.S assembles for gfx950 on its own and tools/asm_hazard_lint.py finds no gfx940+ data hazard in it (the
    assembler inserts no wait states into hand-written code)."""
    import subprocess
    import sys
    # both builds of the loop, then the build-time switches kept for A/B and bring-up (serial symbol fetch; in-loop timers):
    # they have their own copies of the lookups / the literal dispatch and must keep assembling
    for defs in ("", "BRX_WIN_SGPR", "BRX_NO_SPEC", "BRX_NO_SPEC BRX_WIN_SGPR", "BRX_PROF", "LDS_TM_LAST BRX_WIN_SGPR"):  # (the last one: level 4's layout)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_hazard_lint.py")], capture_output=True, text=True,
                           env=dict(os.environ, ASM_DEFS=defs))
        assert r.returncode == 0, r.stdout + r.stderr
        assert "0 finding(s)" in r.stdout
    # the code-length symbol loop of the header path (an asm statement with operands: stand-in registers for the lint)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_hazard_lint.py"), os.path.join(ROOT, "brotli-rs_amd", "csrc", "brx_lens.S")],
                       capture_output=True, text=True, env=dict(os.environ, ASM_DEFS="LDS_LENS=8960"))
    assert r.returncode == 0 and "0 finding(s)" in r.stdout, r.stdout + r.stderr


def test_python_option_numbers_are_the_header_s():
    """brx.OPTIONS (ctypes side) names exactly the BRX_OPTION_* values of include/brx.h."""
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "brx.h")).read()
    enum = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"^\s+BRX_OPTION_([A-Z0-9_]+)\s*=\s*(\d+),?\s", hdr, flags=re.M)}
    assert enum == brx.OPTIONS, (enum, brx.OPTIONS)
