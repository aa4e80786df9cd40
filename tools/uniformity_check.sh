#!/bin/bash
# Compile the kernel to LLVM IR and report, per out-of-line segment, how many loop-header phis LLVM's
# uniformity analysis considers DIVERGENT (wave-uniform decoder state must stay in SGPRs).
set -e
T=${TMPDIR:-/tmp}/brx_uni; mkdir -p $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S -emit-llvm --cuda-device-only -Wno-comment \
   -I brotli-rs_amd/csrc brotli-rs_amd/csrc/brx_kernels.hip -o $T/k.ll 2>/dev/null
/opt/rocm/lib/llvm/bin/opt -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -passes='print<uniformity>' -disable-output $T/k.ll 2> $T/uni.txt
python3 - $T/uni.txt <<'PY'
import re,sys
txt=open(sys.argv[1]).read()
for fn in re.split(r"UniformityInfo for function ", txt)[1:]:
    name=fn.split("\n")[0].strip("':")
    blocks=re.split(r"\nBLOCK ", fn)
    best=None
    for b in blocks[1:]:
        phis=[l for l in b.split("\n") if " = phi " in l]
        if best is None or len(phis)>len(best[1]): best=(b.split("\n")[0],phis)
    if best is None: print('%-45s (no blocks)' % name); continue
    div=[l for l in best[1] if l.strip().startswith("DIVERGENT")]
    tot=sum(1 for l in fn.split("\n") if " br i1 " in l); dv=sum(1 for l in fn.split("\n") if " br i1 " in l and "DIVERGENT" in l)
    print("%-45s biggest loop header BLOCK %s: %d phis, %d divergent; branches %d, divergent %d"%(name,best[0],len(best[1]),len(div),tot,dv))
PY
