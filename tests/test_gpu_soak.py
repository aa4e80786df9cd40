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
# round 6: the bounded / pulled reader (take-backs with the ring and the flush cursor, room for whole meta-blocks, cut and corrupted sources)
SECTIONS += [({}, "reader_fuzz", ["5", "43"]), ({}, "reader_fuzz", ["5", "44"])]
# ... and the node entry over virtual ranks: random batches / capacities / deals / ranks / roots, host and device pointers, RCCL to itself
SECTIONS += [({}, "node_fuzz", ["30", "43"])]
# ... and what a slot holds when its stream fails, against the oracle (odd capacities: every slot alignment): seed 3 is the one that finds
# seg_resume's first-unit hole when the fix is compiled out (profiles/r06_prefix_fuzz.txt)
SECTIONS += [({}, "prefix_fuzz", ["30", "3"]), ({}, "prefix_fuzz", ["30", "43"])]


def _needs_encoder(tool):
    return tool in ("wide_fuzz", "big_fuzz", "small_fuzz", "device_fuzz")  # (reader_fuzz / node_fuzz / prefix_fuzz / gen_fuzz make or hold their streams)


def _section_id(sec):
    e, t, a = sec
    return "%s-%s-%s" % ("+".join("%s=%s" % kv for kv in e.items()), t, "_".join(a))


# Round 6 (788 s of the driver's 1 200 s): the sections are independent processes whose time is mostly the host's
# (encoder, oracle), so they run THREE AT A TIME -- longest first, so that a group's members take about equally long.  Same sections,
# same seeds, same assertions per section.
_COST = {"wide_fuzz": 50, "big_fuzz": 18, "gen_fuzz": 3, "small_fuzz": 9, "device_fuzz": 5, "reader_fuzz": 8, "node_fuzz": 1, "prefix_fuzz": 1}
_ORDERED = sorted(SECTIONS, key=lambda sec: -_COST.get(sec[1], 10) * int(sec[2][0]))
GROUPS = [_ORDERED[i:i + 3] for i in range(0, len(_ORDERED), 3)]


@pytest.mark.parametrize("group", GROUPS, ids=["+".join(_section_id(sec) for sec in g) for g in GROUPS])
def test_soak_sections(group):
    have_enc = True
    if any(_needs_encoder(t) for _, t, _ in group):
        import brotli_enc
        have_enc = brotli_enc.available()
    procs = []
    for env, tool, args in group:
        if _needs_encoder(tool) and not have_enc:
            continue
        full = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env)
        procs.append(((env, tool, args), subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", tool + ".py")] + args, env=full,
                                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    if not procs:
        pytest.skip("libbrotlienc is not in this image")
    failed = []
    for sec, p in procs:
        try:
            out, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            out, err = p.communicate()
            failed.append((_section_id(sec), "TIMEOUT", out[-800:] + err[-800:]))
            continue
        tail = out[-1500:] + err[-1500:]
        if p.returncode != 0 or "MISMATCH" in out or "mismatches" not in out.lower():  # (the fuzzer got as far as its summary line)
            failed.append((_section_id(sec), p.returncode, tail))
    assert not failed, failed


KNOBS = ["BRX_PLAN_B=1", "BRX_PLAN_A=1", "BRX_LOOP_BUILD=1"]


@pytest.mark.timeout(1200)
def test_parity_suite_again_under_forced_knobs():
    """tests/test_gpu_parity.py once more per knob, in fresh processes, with one A/B knob forced on every context the suite makes through
    tests/brx_knobs.py: launch plan B (classification pre-pass, all instances resident at once) on EVERY launch -- plan A decides by
    the context's history --, plan A only, and the sparse-launch build of the assembly loop (the tree cache of many-tree meta-blocks,
    bit window in SGPRs) on full launches too.  Round 4 ran these by hand (profiles/r04_suite_forced.txt); round 6 runs the three
    at the same time (BRX_SUITE_CONCURRENT=1 tells the few tests that assert on kernel TIMES to leave that assertion out)."""
    procs = []
    # (the launch plan never reaches the resumable decode of a reader -- one stream, its own launch: the plan knobs' re-runs leave the
    # reader tests out, the loop-build knob's keeps them)
    readers = "not pulled_reader and not bounded and not every_vector and not taken_back and not makes_room and not format_errors and not streaming_mode"
    for knob in KNOBS:
        name, value = knob.split("=")
        full = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BRX_SUITE_CONCURRENT="1", **{name: value})
        sel = ["-k", readers] if name.startswith("BRX_PLAN") else []
        procs.append((knob, subprocess.Popen([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                                              "-p", "no:cacheprovider"] + sel, env=full, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)))
    failed = []
    for knob, p in procs:
        try:
            out, err = p.communicate(timeout=1100)
        except subprocess.TimeoutExpired:
            p.kill()
            out, err = p.communicate()
            failed.append((knob, "TIMEOUT", out[-1500:]))
            continue
        if p.returncode != 0:
            # (three suites share the GPU here: a failure is taken seriously only if the knob's suite also fails ALONE)
            name, value = knob.split("=")
            alone = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"],
                                   env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **{name: value}), capture_output=True, text=True, timeout=1000, cwd=ROOT)
            if alone.returncode == 0:
                try:
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    open(os.path.join(ROOT, "gpurun_out", "forced_knob_%s_concurrent_only.log" % knob.replace("=", "_")), "w").write(out + "\n---- stderr ----\n" + err)
                except OSError:
                    pass
                continue
            out, err = alone.stdout, alone.stderr
            failed.append((knob, p.returncode, out[-3000:] + err[-1000:]))
            try:  # (the assertion's repr cuts the text short: the whole log goes where gpurun brings it back from)
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                open(os.path.join(ROOT, "gpurun_out", "forced_knob_%s.log" % knob.replace("=", "_")), "w").write(out + "\n---- stderr ----\n" + err)
            except OSError:
                pass
    assert not failed, failed
