"""CPU tests of host-side logic that needs no GPU: launch plan B's grid arithmetic (brotli-rs_amd/csrc/brx_plan.h)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_b_grids_fit_the_chip_and_cover_every_class(tmp_path):
    """brx_plan_b over a sweep of class counts (130 000 cases): every class with streams is decoded by exactly one kernel, no
    grid exceeds its streams or its instance's residency, what is launched together fits the CUs' 40-KiB LDS parts, a mixed
    batch never runs level 1 on its own, the regular class keeps a quarter of the chip; the two measured batches of DESIGN.md
    section 5 get the grids quoted there."""
    exe = str(tmp_path / "plan_test")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "plan_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "0 failures" in out.stdout, out.stdout[-2000:]
