#!/usr/bin/env python3
"""Command census of a stream through the oracle's BRO_TRACE aid: near / far / dictionary copies, commands with
literals, run lengths of consecutive far copies, distance histogram vs candidate ring sizes."""
import collections
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = "import sys; sys.path.insert(0, %r); import oracle_py; oracle_py.decode(open(sys.argv[1],'rb').read())" % os.path.join(ROOT, "tests")
for f in sys.argv[1:]:
    out = subprocess.run([sys.executable, "-c", CODE, f], env=dict(os.environ, BRO_TRACE="1"), capture_output=True, text=True).stderr
    cmds = [tuple(map(int, l.split()[1:])) for l in out.splitlines() if l.startswith("CMD")]  # pos, insert, copy, distance (0 = dictionary)
    far = sum(1 for c in cmds if c[3] > 4096)
    near = sum(1 for c in cmds if 0 < c[3] <= 4096)
    runs, r = collections.Counter(), 0
    for c in cmds:
        isfar = c[3] > 4096
        if isfar and c[1] == 0:
            r += 1
        else:
            if r:
                runs[r] += 1
            r = 1 if isfar else 0
    if r:
        runs[r] += 1
    print(f)
    print("  commands %d: far %d, near %d, dictionary %d; with literals %d; copy > 64 B %d; overlapping %d"
          % (len(cmds), far, near, sum(1 for c in cmds if c[3] == 0), sum(1 for c in cmds if c[1] > 0),
             sum(1 for c in cmds if c[2] > 64), sum(1 for c in cmds if 0 < c[3] < c[2])))
    print("  far-run lengths", sorted(runs.items())[:10])
    print("  copies with distance <= 8K %d, 16K %d, 32K %d, 64K %d" % tuple(sum(1 for c in cmds if 0 < c[3] <= k) for k in (8192, 16384, 32768, 65536)))
