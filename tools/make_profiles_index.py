#!/usr/bin/env python3
"""Write profiles/INDEX.md: the hand-kept table "which file backs which number of DESIGN.md" (tools/profiles_index_head.md) followed by
a generated listing of every file under profiles/ with what kind of evidence it is (by name pattern)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
KINDS = [
    (r"bench_(r\d+[a-z]?)_(.+)_under_rocprof\.json", "bench.py line of `{1}` while rocprofv3 --kernel-trace --stats ran (pairs with `{0}_{1}_kernel_stats.csv`)"),
    (r"bench_(r\d+[a-z]?)_default.*\.json", "the default `python bench.py` line (what the driver runs): value, roofline incl. in-run PMC traffic, cpu_baseline, copy_path"),
    (r"bench_(r\d+[a-z]?)_(.+)\.json", "bench.py line of workload `{1}` (kernel ms by HIP events, chain floor, bit_exact)"),
    (r"(r\d+[a-z]?)_(.+)_kernel_stats\.csv", "rocprofv3 --kernel-trace --stats summary of `bench.py --workload {1}`: calls, total / average / min / max ns per kernel"),
    (r"(r\d+[a-z]?)_pmc.*\.(txt|csv)", "rocprofv3 --pmc SQ counters per dispatch (instructions by class, busy / wait cycles)"),
    (r"(r\d+[a-z]?)_sweep\.txt", "streams-per-GPU sweeps: MB/s and kernel ms from 1 to 16 384 streams"),
    (r"(r\d+[a-z]?)_pcie\.txt", "the host-pointer path, PCIe inclusive (pinned in place / pageable)"),
    (r"(r\d+[a-z]?)_residency\.txt", "per-stream start / end / CU traces of mixed batches (BRX_OPTION_TRACE)"),
    (r"(r\d+[a-z]?)_soak.*\.txt", "differential soak logs (fuzzers vs oracle), builder-run"),
    (r"(r\d+[a-z]?)_phases.*\.txt", "per-phase cycle timers of single streams (BRX_BRINGUP build)"),
    (r"hbm_traffic\.json", "HBM bytes per launch per workload from FETCH_SIZE / WRITE_SIZE passes, stamped with the kernel source id bench.py checks"),
    (r"EXPERIMENTS\.md", "the record of rounds 1 - 4: superseded numbers, everything measured and dropped"),
    (r"INDEX\.md", "this file"),
]


def describe(name):
    for pat, text in KINDS:
        m = re.fullmatch(pat, name)
        if m:
            return text.format(*m.groups())
    return None


def main():
    head = open(os.path.join(ROOT, "tools", "profiles_index_head.md")).read()
    names = sorted(os.listdir(P))
    rounds = {}
    for n in names:
        m = re.search(r"(?:^|_)(r0\d)", n)
        rounds.setdefault(m.group(1) if m else "other", []).append(n)
    out = [head.rstrip(), "", "## Every file", ""]
    for r in sorted(rounds, reverse=True):
        out.append("### %s" % r)
        out.append("")
        for n in rounds[r]:
            d = describe(n)
            if d is None:
                first = ""
                try:
                    with open(os.path.join(P, n), errors="replace") as f:
                        for line in f:
                            line = line.strip().lstrip("#/ ").strip()
                            if line:
                                first = line[:160]
                                break
                except OSError:
                    pass
                d = "experiment log -- " + first if first else "experiment log"
            out.append("* `%s` — %s" % (n, d))
        out.append("")
    open(os.path.join(P, "INDEX.md"), "w").write("\n".join(out) + "\n")
    print("profiles/INDEX.md: %d files" % len(names))


if __name__ == "__main__":
    main()
