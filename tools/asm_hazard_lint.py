#!/usr/bin/env python3
"""Wait-state lint for the hand-written command loop (brotli-rs_amd/csrc/brx_hot.S).

The assembler inserts no wait states into hand-written code.  This script assembles the loop for gfx950, walks the
disassembly in fall-through order and reports the gfx940+/gfx950 data hazards that bit during bring-up:

  R1  VALU writes an SGPR (v_readlane / v_readfirstlane / v_cmp ... into s#)  ->  a VALU reads it as an operand: 2 wait states
  R2  VALU writes an SGPR                                                   ->  v_readlane / v_writelane lane select: 4
  R3  VALU writes an SGPR                                                   ->  VMEM uses it (address / resource): 5
  R4  VALU writes a VGPR                                                    ->  v_readlane / v_readfirstlane reads it: 1
  R5  VALU writes a VGPR                                                    ->  a DPP instruction reads it: 2

A wait state = any instruction issued in between (s_nop N counts N + 1).  Sequences are cut at unconditional control
transfers; paths that ENTER a sequence through a taken branch are not modelled (the branch itself costs wait states).
Exit code 1 when something is found.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "brotli-rs_amd", "csrc", "brx_hot.S")
CLANG = "/opt/rocm/lib/llvm/bin/clang"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble():
    with tempfile.TemporaryDirectory() as t:
        pp, obj = os.path.join(t, "hot.s"), os.path.join(t, "hot.o")
        subprocess.check_call(["cpp", "-P", "-x", "assembler-with-cpp"] + ["-D" + d for d in os.environ.get("ASM_DEFS", "").split()] + [SRC, "-o", pp])
        txt = open(pp).read()
        if "@" in txt:  # an asm statement with operands (brx_lens.S): `@n@` = operand n -- stand-in registers for the lint
            ops = {0: "s0", 1: "s1", 2: "s2", 3: "s3", 4: "s[4:5]", 5: "s6", 6: "s7", 7: "s8", 8: "v8", 9: "v9", 10: "v10", 11: "v11",
                   12: "v12", 13: "s13", 14: "s14", 15: "s15", 16: "s16", 17: "s[18:19]", 18: "v18", 19: "v19"}
            txt = re.sub(r"@(\d+)@", lambda m: ops[int(m.group(1))], txt)
            open(pp, "w").write(txt)
        subprocess.check_call([CLANG, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", pp, "-o", obj])
        out = subprocess.check_output([OBJDUMP, "-d", obj]).decode()
    ins = []
    for line in out.splitlines():
        m = re.match(r"\s+([a-z_0-9]+)\s*(.*?)\s*//", line)
        if m:
            ins.append((m.group(1), m.group(2)))
    return ins


def regs(tok, kind):
    """register numbers of class `kind` ('s' or 'v') named by one operand token"""
    tok = tok.strip()
    m = re.fullmatch(kind + r"(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(kind + r"\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    if kind == "s" and tok in ("vcc", "vcc_lo", "vcc_hi"):
        return {"vcc"}
    return set()


def main():
    ins = disassemble()
    last_sgpr = {}  # sgpr -> index of the VALU instruction that wrote it
    last_vgpr = {}  # vgpr -> index of the VALU instruction that wrote it
    pos = 0         # issue slots so far (s_nop N advances N + 1)
    found = []
    for n, (op, args) in enumerate(ins):
        base = re.sub(r"_e(32|64)$", "", op)
        toks = [a for a in re.split(r",\s*", args) if a] if args else []
        toks = [t.split(" ")[0] for t in toks]
        valu = base.startswith("v_")
        vmem = base.startswith(("buffer_", "global_", "scratch_", "flat_"))
        dst, srcs = (toks[0], toks[1:]) if toks else ("", [])
        if base.startswith(("ds_write", "buffer_store", "global_store", "s_cmp", "s_bitcmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_setpc")):
            dst, srcs = "", toks
        if valu:
            for t in srcs:
                for r in regs(t, "s"):
                    if r in last_sgpr and r != "vcc":
                        gap = pos - last_sgpr[r] - 1
                        lane_sel = base in ("v_readlane_b32", "v_writelane_b32") and t == srcs[-1]
                        need = 4 if lane_sel else 2
                        if gap < need:
                            found.append("%s: #%d %s %s reads s%s written by a VALU %d slot(s) earlier (needs %d between)"
                                         % ("R2" if lane_sel else "R1", n, op, args, r, gap + 1, need))
            if base in ("v_readlane_b32", "v_readfirstlane_b32"):
                for r in regs(srcs[0], "v"):
                    if r in last_vgpr and pos - last_vgpr[r] - 1 < 1:
                        found.append("R4: #%d %s %s reads v%d written by the previous VALU instruction" % (n, op, args, r))
            if base.endswith("_dpp"):
                for t in srcs:
                    for r in regs(t, "v"):
                        if r in last_vgpr and pos - last_vgpr[r] - 1 < 2:
                            found.append("R5: #%d %s %s reads v%d through DPP %d slot(s) after a VALU wrote it (needs 2 between)"
                                         % (n, op, args, r, pos - last_vgpr[r]))
        if vmem:
            for t in toks:
                for r in regs(t, "s"):
                    if r in last_sgpr and r != "vcc" and pos - last_sgpr[r] - 1 < 5:
                        found.append("R3: #%d %s %s uses s%s written by a VALU %d slot(s) earlier (needs 5 between)"
                                     % (n, op, args, r, pos - last_sgpr[r]))
        # bookkeeping of writers
        if valu:
            if base in ("v_readlane_b32", "v_readfirstlane_b32") or (base.startswith("v_cmp") and regs(dst, "s")):
                for r in regs(dst, "s"):
                    last_sgpr[r] = pos
            else:
                for r in regs(dst, "v"):
                    last_vgpr[r] = pos
        elif not vmem and not base.startswith("ds_"):
            for r in regs(dst, "s"):  # a scalar write supersedes an older VALU write
                last_sgpr.pop(r, None)
        if base.startswith(("ds_read", "buffer_load", "global_load")):
            for r in regs(dst, "v"):
                last_vgpr.pop(r, None)
        pos += 1
        if base == "s_nop":
            pos += int(args.strip() or 0)
        if base in ("s_branch", "s_setpc_b64"):
            last_sgpr.clear()
            last_vgpr.clear()
    print("%d instructions checked, %d finding(s)" % (len(ins), len(found)))
    for f in found:
        print("  " + f)
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
