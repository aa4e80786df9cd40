#!/usr/bin/env python3
"""Static instruction census of the compiled (C++) segments of the decode kernel: instructions, scalar / vector / branch
split and SGPR spill traffic (v_writelane / v_readlane into the callee-saved VGPRs) per function.
usage: asm_census.py [file.s ...]   (no argument: compiles brotli-rs_amd/csrc/brx_kernels.hip for gfx950 to /tmp first)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def compile_to(path, defs=()):
    src = os.path.join(ROOT, "brotli-rs_amd", "csrc", "brx_kernels.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-comment",
                           "-Wno-inline-asm", *defs, src, "-o", path], cwd=os.path.dirname(src), stderr=subprocess.DEVNULL)

def census(path):
    funcs, cur = {}, None
    for l in open(path):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            cur = m.group(1)
            funcs[cur] = dict(n=0, salu=0, valu=0, br=0, lds=0, vmem=0, spill_w=0, spill_r=0)
        elif l.startswith('\t.size'):
            cur = None
        elif cur and l.startswith('\t') and not l.strip().startswith(('.', ';')):
            parts = l.split()
            if not parts:
                continue
            op, f = parts[0], funcs[cur]
            f['n'] += 1
            if op.startswith('s_cbranch') or op in ('s_branch', 's_setpc_b64', 's_swappc_b64'): f['br'] += 1
            elif op.startswith('s_'): f['salu'] += 1
            elif op.startswith('v_'): f['valu'] += 1
            elif op.startswith('ds_'): f['lds'] += 1
            else: f['vmem'] += 1
            if re.match(r'\tv_writelane_b32 v4[0-7]', l): f['spill_w'] += 1
            if re.match(r'\tv_readlane_b32 s\d+, v4[0-7]', l): f['spill_r'] += 1
    return funcs

if __name__ == "__main__":
    paths = sys.argv[1:]
    if not paths:
        compile_to("/tmp/brx_k.s")
        paths = ["/tmp/brx_k.s"]
    for p in paths:
        print(p)
        for k, v in census(p).items():
            if v['n'] > 50:
                print("  %-48s" % k, " ".join("%s=%d" % kv for kv in v.items()))
