"""bench.py's own launcher: `python bench.py --gpus N` must start N ranks by itself, and must refuse
to print a number when it cannot.  CPU only: BRX_BENCH_STUB=1 swaps the decode for a stand-in and RCCL for gloo -- what is
tested is the launch, the rendezvous, the max-over-ranks reduction and the shape of the line, never a rate."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, stub):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    if stub:
        env["BRX_BENCH_STUB"] = "1"
    else:
        env.pop("BRX_BENCH_STUB", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=600)


def test_gpus_2_starts_two_ranks():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "monkeyx16384", "--streams", "8"], stub=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["stub"] and res["data"] == "STUB"
    assert len(res["kernel_ms_per_rank"]) == 2
    assert res["strong"]["streams_per_gpu"] == 4 and res["strong"]["streams_total"] == 8
    assert res["scaling"] == "weak" and res["bit_exact"] is True


def test_gpus_2_without_two_gpus_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs present")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], stub=False)
    assert r.returncode != 0
    assert not any(l.startswith("{") for l in r.stdout.splitlines()), r.stdout
    assert "refusing" in r.stderr


def test_world_size_mismatch_fails_loudly():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", BRX_BENCH_STUB="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not any(l.startswith("{") for l in r.stdout.splitlines())
