"""Build libbrx.so (HIP kernels + C ABI) in-tree for gfx950.  hipcc cross-compiles without a GPU."""
import os
import re
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_PATH = os.path.join(PKG, "libbrx.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["brx_kernels.hip", "brx_kernels_l1.hip", "brx_kernels_l2.hip", "brx_kernels_l3.hip", "brx_kernels_l4.hip", "brx_kernels_s.hip", "brx_gen.hip", "brx_util.hip", "brx_api.cpp", "brx_node.cpp"]
DEPS = SOURCES + ["brx_device.h", "brx_internal.h", "brx_plan.h", "brx_small.h", "brx_hot.S", "brx_lens.S", os.path.join("..", "host", "brx_walk.cpp"), os.path.join("..", "..", "include", "brx.h"),
                  os.path.join("..", "tables", "dictionary.bin"), os.path.join("..", "tables", "context_lut.bin"),
                  os.path.join("..", "tables", "transforms.bin"), os.path.join("..", "tables", "gen_header.bin"), os.path.join("..", "build.py")]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build_library(force=False, verbose=False):
    """Compile csrc/ into brotli-rs_amd/libbrx.so.  Returns the library path.  Safe to call from several processes at
    once (one rank per GPU): the build is serialised by a lock file and the library is replaced atomically."""
    if not force and not _stale():
        return LIB_PATH
    import fcntl
    with open(os.path.join(PKG, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():  # another process built it while this one waited
            return LIB_PATH
        return _build_locked(verbose)


def _build_locked(verbose):
    if not os.path.exists(HIPCC):
        if os.path.exists(LIB_PATH):
            return LIB_PATH  # GPU box without a need to rebuild: use the prebuilt library that travelled
        raise RuntimeError("hipcc not found at %s and no prebuilt libbrx.so" % HIPCC)
    gen = os.path.join(CSRC, "_gen", "brx_tables_gen.h")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "bin2h.py"), gen, "BRX", "static const"])
    # the hand-written command loop: cpp resolves the register names, the text becomes one asm statement
    prof = ["-DBRX_PROF"] if os.environ.get("BRX_PROF") == "1" else []  # bring-up: timers inside the loop
    if os.environ.get("BRX_NO_SPEC") == "1":
        prof.append("-DBRX_NO_SPEC")  # A/B: serial symbol fetch instead of the lane-speculative one
    if os.environ.get("BRX_PIN_NOPS"):
        prof.append("-DPIN_NOPS=%d" % int(os.environ["BRX_PIN_NOPS"]))  # A/B: position of the loop (profiles/r03_pins.txt)
    extra = ["-D" + d for d in os.environ.get("BRX_DEFS", "").split()]  # A/B: defines for the loop AND the C++ side
    prof += extra
    # two builds of the loop (brx_hot.S, "Two builds of this file"): bit window in VGPRs (full chip) / in SGPRs (few waves per CU)
    # ... each for the four instances of the kernel (brx_device.h: the wider ones have their LDS offsets LDS_GROW further up)
    variants = [("brx_hot_asm.h", [], None), ("brx_hot_asm_sw.h", ["-DBRX_WIN_SGPR"], ".LS_")]
    for level, grow in ((1, 2560), (2, 10240), (3, 30720)):
        variants += [("brx_hot_asm_l%d.h" % level, ["-DLDS_GROW=%d" % grow], None),
                     ("brx_hot_asm_sw_l%d.h" % level, ["-DLDS_GROW=%d" % grow, "-DBRX_WIN_SGPR"], ".LS_")]
    # level 4 (150 KiB of LDS, one workgroup per CU): the table memory last, nothing else moves with its size
    variants += [("brx_hot_asm_l4.h", ["-DLDS_TM_LAST"], None), ("brx_hot_asm_sw_l4.h", ["-DLDS_TM_LAST", "-DBRX_WIN_SGPR"], ".LS_")]
    for name, defs, prefix in variants:
        hot = subprocess.check_output(["cpp", "-P", "-x", "assembler-with-cpp"] + prof + defs + [os.path.join(CSRC, "brx_hot.S")]).decode()
        assert ")BRXASM" not in hot and "%" not in hot and "{" not in hot and "$" not in hot
        if prefix:
            hot = hot.replace(".L", prefix)  # both texts land in one assembly file: distinct local labels
        macros = re.findall(r"^\s*\.macro\s+(\w+)", hot, flags=re.M)
        hot += "".join(".purgem %s\n" % m for m in macros)  # ... and macro names free again after each
        with open(os.path.join(CSRC, "_gen", name), "w") as f:
            f.write("// generated from brx_hot.S by build.py -- do not edit\n")
            f.write('R"BRXASM(\n' + hot + ')BRXASM"\n')
    # the code-length symbol loop of the header path (brx_lens.S): one asm statement WITH operands -- `@n@` in the source is
    # operand n, local labels get the statement's unique suffix
    # (LDS_LENS = offset of Lds::lens: behind the ring and the table memory; "s" = the lean instance, brx_small.h)
    for level, lens_at in ((0, 8960), (1, 8960 + 2560), (2, 8960 + 10240), (3, 8960 + 30720), (4, 2048), ("s", 2048 + 2048)):
        txt = subprocess.check_output(["cpp", "-P", "-x", "assembler-with-cpp", "-DLDS_LENS=%d" % lens_at,
                                       os.path.join(CSRC, "brx_lens.S")]).decode()
        assert ")BRXASM" not in txt and "%" not in txt and "{" not in txt and "$" not in txt
        txt = re.sub(r"@(\d+)@", r"%\1", txt).replace(".Lls_", ".Lls%=_")
        with open(os.path.join(CSRC, "_gen", "brx_lens_asm%s.h" % ("_s" if level == "s" else "_l%d" % level if level else "")), "w") as f:
            f.write("// generated from brx_lens.S by build.py -- do not edit\n")
            f.write('R"BRXASM(\n' + txt + ')BRXASM"\n')
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-comment"]
    if os.environ.get("BRX_BRINGUP") == "1" or os.environ.get("BRX_PROF") == "1":
        cmd.append("-DBRX_BRINGUP")  # bring-up: BRX_DEBUG_STATS / BRX_DEBUG_STOP=9 + BRX_DEBUG_DUMP (tools/gpu_dumps.sh, tools/span_stats.py)
    cmd += extra + [os.path.join(CSRC, s) for s in SOURCES]
    tmp = "%s.tmp.%d" % (LIB_PATH, os.getpid())
    cmd += ["-ldl", "-lpthread", "-o", tmp]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(tmp, LIB_PATH)
    # host program above the C ABI: the reference's file walker (src/main.rs:49-70) on the batched decoder
    walk = os.path.join(PKG, "brx_walk")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(PKG, "host", "brx_walk.cpp"), "-o", walk + ".tmp", "-L", PKG,
                           "-lbrx", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    os.replace(walk + ".tmp", walk)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
