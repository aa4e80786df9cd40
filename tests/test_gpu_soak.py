"""The differential soak in the driver-run suite.  Round 4's status-0-wrong-bytes bug (late resume with an unaligned
output slot, c8bba1d) was found by tools/wide_fuzz.py on the builder's metered GPU, not by `pytest -m gpu`: the fuzzers of
tools/gpu_soak_r04.sh run here as fixed-seed, time-boxed sections -- HIP path against the oracle / the original bytes, both launch
plans, both builds of the assembly loop, a small slab pool.  Reference practice mirrored: the AFL findings of
/root/reference/docs/notes_afl.txt:14-20 became the regression tests of tests/lib.rs:379-605.

Each section is one fresh process (the A/B knobs reach the library through tests/brx_knobs.py from the process environment; the
library itself reads none).  A section fails on any MISMATCH line or a non-zero exit."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

# (environment, tool, arguments: rounds seed [late]) -- seed 43 is the one that found c8bba1d
PLANS = ["BRX_PLAN_A", "BRX_PLAN_B"]
FUZZERS = [("wide_fuzz", ["2", "43", "late"]), ("wide_fuzz", ["1", "44"]), ("big_fuzz", ["2", "43"]), ("gen_fuzz", ["4", "43"]),
           ("small_fuzz", ["2", "43"]), ("device_fuzz", ["4", "43"])]
SECTIONS = [({p: "1"}, t, a) for p in PLANS for t, a in FUZZERS]
# round 5's extended soak found this one (pre-existing): 199 slab-class streams in a catch-all launch whose grid ignored BRX_GRID_CAP
# while the slab pool followed it -- waiters gave up after 0.5 s and valid streams came back with status 27
SECTIONS += [({p: "1"}, "device_fuzz", ["3", "302"]) for p in PLANS]
SECTIONS += [({"BRX_LOOP_BUILD": str(b)}, "wide_fuzz", ["1", "45"]) for b in (0, 1)]
SECTIONS += [({"BRX_LOOP_BUILD": str(b)}, "gen_fuzz", ["2", "46"]) for b in (0, 1)]
SECTIONS += [({"BRX_GRID_CAP": "64"}, "wide_fuzz", ["2", "47"])]  # a pool of 64 slabs under corrupted wide streams: a slab not given back stalls this
# round 5: every tenth stream ONE piece of 1.2 .. 2.5 MiB of text + an ELF image -- more than 64 trees of a kind, tables for the level-4
# instance -- and their corrupted / truncated variants, under both plans
SECTIONS += [({p: "1"}, "wide_fuzz", ["1", "611", "big"]) for p in PLANS]
SECTIONS += [({"BRX_DEBUG_STOP": "8"}, "big_fuzz", ["1", "48"])]  # the C++ command loops only (the safety net of the assembly loop)


def _needs_encoder(tool):
    return tool in ("wide_fuzz", "big_fuzz", "small_fuzz", "device_fuzz")


@pytest.mark.parametrize("env,tool,args", SECTIONS, ids=["%s-%s-%s" % ("+".join("%s=%s" % kv for kv in e.items()), t, "_".join(a)) for e, t, a in SECTIONS])
def test_soak_section(env, tool, args):
    if _needs_encoder(tool):
        import brotli_enc
        if not brotli_enc.available():
            pytest.skip("libbrotlienc is not in this image")
    full = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool + ".py")] + args, env=full, capture_output=True, text=True, timeout=600)
    tail = (r.stdout[-1500:] + r.stderr[-1500:])
    assert r.returncode == 0 and "MISMATCH" not in r.stdout, tail
    assert "mismatches" in r.stdout.lower(), tail  # (the fuzzer got as far as its summary line)


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("knob", ["BRX_PLAN_B=1", "BRX_PLAN_A=1", "BRX_LOOP_BUILD=1"])
def test_parity_suite_again_under_a_forced_knob(knob):
    """tests/test_gpu_parity.py once more, in a fresh process, with one A/B knob forced on every context the suite makes through
    tests/brx_knobs.py: launch plan B (classification pre-pass, all instances resident at once) on EVERY launch -- plan A decides by
    the context's history --, plan A only, and the sparse-launch build of the assembly loop (the tree cache of many-tree meta-blocks,
    bit window in SGPRs) on full launches too.  Round 4 ran these by hand (profiles/r04_suite_forced.txt)."""
    name, value = knob.split("=")
    full = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **{name: value})
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                        "-p", "no:cacheprovider"], env=full, capture_output=True, text=True, timeout=1100, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
