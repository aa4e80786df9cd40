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
    assert len(res["strong"]["kernel_ms_per_rank"]) == 2 and res["strong"]["unit"] == "MB/s" and "what" in res["strong"]
    assert res["steps"] == 2 and res["warmup"] == 1 and res["higher_is_better"] is True and res["vs_baseline"] is None
    assert "poisoned" in res["bit_exact_of"]


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


def test_in_run_traffic_reads_the_pmc_passes(tmp_path, monkeypatch):
    """roofline.traffic is measured by the bench itself: three child passes under `rocprofv3 --pmc` (bench.measured_traffic).  Here
    a stand-in `rocprofv3` on PATH writes the counter files a real pass leaves; what is tested is the command line (one counter
    per pass, --kernel-trace, no other trace domain, the child told not to recurse), the per-launch sum over the regular
    kernel and the wider instances behind it, the KiB unit and the calibrated x 2 of FETCH_SIZE."""
    fake = tmp_path / "rocprofv3"
    fake.write_text('''#!%s
import os, sys
a = sys.argv[1:]
d = a[a.index("-d") + 1]
i = a.index("--pmc") + 1
cs = []
while not a[i].startswith("--"):
    cs.append(a[i]); i += 1
assert "--kernel-trace" in a and "--no-traffic" in a and "--no-chain-floor" in a and "--sys-trace" not in a and "-s" not in a and "--hip-trace" not in a
assert a.count("--pmc") == 1 and (len(cs) == 1 or all(c.startswith("SQ_") for c in cs))  # FETCH_SIZE / WRITE_SIZE: a pass each
open(os.path.join(os.environ["FAKE_LOG"]), "a").write("+".join(cs) + "\\n")
os.makedirs(os.path.join(d, "host", "123"), exist_ok=True)
rows = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value"]
for c in cs:
    v = {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 300.0, "SQ_INSTS_SALU": 7000.0, "SQ_INSTS_VALU": 5000.0, "SQ_BUSY_CYCLES": 3200.0}[c]
    for disp in (1, 5):  # two launches: the regular kernel + one wider instance behind each, and a kernel that is not ours
        rows.append('%%d,"brx_decode_kernel(BrxKernelArgs)",%%s,%%f' %% (disp, c, v))
        rows.append('%%d,"brx_decode_kernel_l3(BrxKernelArgs)",%%s,%%f' %% (disp + 1, c, 24.0))
        rows.append('%%d,"fill_kernel",%%s,%%f' %% (disp + 2, c, 1e9))
open(os.path.join(d, "host", "123", "p_counter_collection.csv"), "w").write("\\n".join(rows) + "\\n")
''' % sys.executable)
    fake.chmod(0o755)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    monkeypatch.setenv("FAKE_LOG", str(tmp_path / "log.txt"))
    sys.path.insert(0, ROOT)
    import bench
    total, detail = bench.measured_traffic("alice29x4096", 0)
    assert (tmp_path / "log.txt").read_text().split() == ["FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_SALU+SQ_INSTS_VALU+SQ_BUSY_CYCLES"]
    assert detail == {"fetch_size_counter_bytes": 1024 * 1024, "fetch_bytes": 2 * 1024 * 1024, "write_bytes": 324 * 1024,
                      "sq": {"SQ_INSTS_SALU": 7024, "SQ_INSTS_VALU": 5024, "SQ_BUSY_CYCLES": 3224}}
    assert total == 2 * 1024 * 1024 + 324 * 1024
    assert not bench.under_profiler()
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", "/tmp/x")
    assert bench.under_profiler()
