#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/ubench/fetchcal.hip) -> gpurun_out/r03_fetchcal.txt
set -u
R=$GRAFT_REPO_ROOT
# (the binary is built here, every time: it is not tracked)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/ubench/fetchcal.hip -o $R/tools/ubench/fetchcal || exit 1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/fc_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/fc_$c -o p -- $R/tools/ubench/fetchcal > /tmp/fc_$c.log 2>&1
done
python3 - <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/r03_fetchcal.txt
import csv, glob
GiB = 2 << 30
expect = {"k_byte_per_line": None, "k_byte_adjacent": GiB, "k_dword_adjacent": GiB, "k_b128_adjacent": GiB, "k_store_b128": GiB}
print("kernel (in launch order)            counter      reported bytes   known bytes touched   ratio")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob("/tmp/fc_%s/**/*counter_collection.csv" % c, recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and r["Kernel_Name"].startswith("k_")]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    seen = {}
    for r in rows:
        name = r["Kernel_Name"].split("(")[0]
        k = seen.get(name, 0); seen[name] = k + 1
        rep = float(r["Counter_Value"]) * 1024.0
        if name == "k_byte_per_line":
            known = GiB if k == 0 else GiB // 2  # lines of 64 B touched: stride 64 -> every line; stride 128 -> every other line
            label = name + ("(stride 64)" if k == 0 else "(stride 128)")
        else:
            known, label = expect[name], name
        if (c == "WRITE_SIZE") != (name == "k_store_b128"):
            continue
        print("%-34s %-11s %16.0f %20d   %.3f" % (label, c, rep, known, rep / known))
PY
