// litloop.hip -- one lone wave, cycles per literal of candidate literal-loop bodies (round 5: why the pipelined four-slot loop
// of brx_hot.S is or is not faster than the per-literal one).  Every body is the real instruction sequence on synthetic tables:
// every code is 5 bits long, the symbol lists live in LDS, the chain (entry -> context -> tree -> compare -> length -> window)
// is the real one.  Build: hipcc --offload-arch=gfx950 -O2 litloop.hip -o litloop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;

// registers: s[60:61] window, s62 SNAV, s63 T4 (context id), s64 T5, s65 T7, s66 loop counter, s67.. scratch, m0
// v10 VSH (32 - lane & 15), v11 limits (counts), v12 bases (folded), v13 VCMAP / VSLOT, v14.. scratch
#define PRE                                                                                                  \
    "v_mbcnt_lo_u32_b32 v20, -1, 0\n v_mbcnt_hi_u32_b32 v20, -1, v20\n"                                      \
    "v_and_b32 v21, 15, v20\n v_sub_u32 v10, 32, v21\n v_min_u32 v10, 31, v10\n"                             \
    "v_mov_b32 v11, 0\n v_cmp_lt_u32 vcc, 4, v21\n v_mov_b32 v22, 1\n v_lshlrev_b32 v22, v21, v22\n"         \
    "v_cndmask_b32 v11, v11, v22, vcc\n"          /* lengths 5.. : limit = 2^L (every code of that length is below it) */ \
    "v_mov_b32 v12, 0\n"                          /* folded base: list at LDS 0 */                            \
    "v_and_b32 v13, 3, v20\n v_lshlrev_b32 v13, 4, v13\n" /* context -> 16 * slot (and, doubled, a tree pair index 0..6 for the base loop) */ \
    "s_mov_b32 s60, 0x9e3779b9\n s_mov_b32 s61, 0x7f4a7c15\n s_mov_b32 s62, 0\n s_mov_b32 s63, 1\n s_mov_b32 s64, 2\n s_mov_b32 s65, 3\n" \
    "s_mov_b32 s66, 1023\n s_mov_b32 m0, 1023\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n"
#define TAKE_S "s_lshr_b64 s[60:61], s[60:61], s68\n s_or_b32 s61, s61, 0x9e370000\n s_sub_u32 s62, s62, s68\n s_cbranch_scc1 9f\n9:\n"

#define KERNEL(NAME, EXEC, BODY)                                                                             \
    __global__ void NAME(u64 *out) {                                                                         \
        __shared__ u32 lds[1024];                                                                            \
        for (u32 i = threadIdx.x; i < 1024; i += 64) lds[i] = ((i * 37u) & 0xffu) | (((i * 5u) & 0xffu) << 8) | ((((i * 11u) & 0xffu) | 0x2000u) << 16); \
        __syncthreads();                                                                                     \
        u64 t0, t1;                                                                                          \
        asm volatile(PRE "s_mov_b64 exec, " EXEC "\n s_memtime %0\n s_waitcnt lgkmcnt(0)\n"                  \
                     BODY                                                                                    \
                     "s_waitcnt lgkmcnt(0)\n s_memtime %1\n s_waitcnt lgkmcnt(0)\n s_mov_b64 exec, -1\n"     \
                     : "=s"(t0), "=s"(t1) : : "vcc", "scc", "memory", "m0", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", \
                       "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", \
                       "v19", "v20", "v21", "v22", "v30", "v31", "v32", "v33", "v9", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", \
                       "v50", "v51", "v52", "v53", "v54");                              \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                                              \
    }

// the per-literal loop of round 4 (LIT_R_BODY_M3, sparse build): tree pair through the VGPR index mode, fetch waited for
KERNEL(k_base, "0x1ffff",
    "1:\n"
    "v_readlane_b32 s67, v13, s63\n"
    "v_bfrev_b32 v14, s60\n v_lshrrev_b32 v15, v10, v14\n"
    "s_lshr_b32 s67, s67, 8\n"                       // (index 0: the one pair there is)
    "s_set_gpr_idx_on s67, 6\n v_cmp_lt_u32 vcc, v15, v11\n v_lshl_add_u32 v15, v15, 1, v12\n s_set_gpr_idx_off\n"
    "ds_read_u16 v16, v15\n"
    "s_ff1_i32_b32 s68, vcc_lo\n"
    TAKE_S
    "s_waitcnt lgkmcnt(0)\n"
    "v_readlane_b32 s69, v16, s68\n"
    "s_bfe_u32 s65, s69, 0x3000d\n s_lshl3_add_u32 s63, s65, s64\n s_and_b32 s63, s63, 63\n"
    "v_mov_b32 v17, s69\n ds_write_b8 v30, v17 offset:2048\n v_add_u32 v30, 1, v30\n"
    "s_sub_u32 s66, s66, 1\n s_cbranch_scc0 1b\n")

// the pipelined four-slot loop as first written (SL_SELECT / SL_COMPARE / SL_FINISH_M3, M0 as the counter)
#define SL_COMPARE(vs, vccp) "v_bfrev_b32 v14, s60\n v_lshrrev_b32 v15, v10, v14\n v_cmp_lt_u32 " vccp ", v15, v11\n v_lshl_add_u32 v15, v15, 1, v12\n ds_read_u16 " vs ", v15\n"
#define SL_SELECT(vccp) "v_readlane_b32 s67, v13, s63\n s_bitcmp1_b32 s67, 7\n s_cbranch_scc1 8f\n s_lshr_b64 s[70:71], " vccp ", s67\n s_ff1_i32_b32 s68, s70\n s_add_u32 s67, s67, s68\n" TAKE_S
#define SL_FINISH(vs, cnt, cur, prev) "s_waitcnt lgkmcnt(" cnt ")\n v_readlane_b32 s69, " vs ", s67\n s_bfe_u32 " cur ", s69, 0x3000d\n s_lshl3_add_u32 s63, " cur ", " prev "\n s_and_b32 s63, s63, 63\n v_writelane_b32 v31, s69, m0\n"
KERNEL(k_slot, "-1",
    SL_COMPARE("v16", "s[72:73]")
    "1:\n"
    SL_SELECT("s[72:73]")
    "s_sub_u32 m0, m0, 1\n s_cbranch_scc1 8f\n"
    SL_COMPARE("v17", "s[74:75]")
    SL_FINISH("v16", "1", "s65", "s64")
    SL_SELECT("s[74:75]")
    "s_sub_u32 m0, m0, 1\n s_cbranch_scc1 8f\n"
    SL_COMPARE("v16", "s[72:73]")
    SL_FINISH("v17", "1", "s64", "s65")
    "s_branch 1b\n8:\n")

// ... the counter in an SGPR, M0 copied for the v_writelane
#define SL_FINISH_B(vs, cnt, cur, prev) "s_waitcnt lgkmcnt(" cnt ")\n v_readlane_b32 s69, " vs ", s67\n s_bfe_u32 " cur ", s69, 0x3000d\n s_lshl3_add_u32 s63, " cur ", " prev "\n s_and_b32 s63, s63, 63\n v_mov_b32 v33, s69\n ds_write_b8 v30, v33 offset:2048\n"
KERNEL(k_slot_dsw, "-1",
    SL_COMPARE("v16", "s[72:73]")
    "1:\n"
    SL_SELECT("s[72:73]")
    "s_sub_u32 s66, s66, 1\n s_cbranch_scc1 8f\n"
    SL_COMPARE("v17", "s[74:75]")
    SL_FINISH_B("v16", "2", "s65", "s64")
    SL_SELECT("s[74:75]")
    "s_sub_u32 s66, s66, 1\n s_cbranch_scc1 8f\n"
    SL_COMPARE("v16", "s[72:73]")
    SL_FINISH_B("v17", "2", "s64", "s65")
    "s_branch 1b\n8:\n")

// ... not pipelined: the same body, the fetch waited for at once (what the overlap is worth)
KERNEL(k_slot_serial, "-1",
    "1:\n"
    SL_COMPARE("v16", "s[72:73]")
    SL_SELECT("s[72:73]")
    SL_FINISH("v16", "0", "s65", "s64")
    "s_sub_u32 m0, m0, 1\n s_cbranch_scc0 1b\n8:\n")

// ... the compare into vcc + s_mov_b64 instead of a VOP3 SGPR destination
#define SL_COMPARE_V(vs, vccp) "v_bfrev_b32 v14, s60\n v_lshrrev_b32 v15, v10, v14\n v_cmp_lt_u32 vcc, v15, v11\n v_lshl_add_u32 v15, v15, 1, v12\n ds_read_u16 " vs ", v15\n s_mov_b64 " vccp ", vcc\n"
KERNEL(k_slot_vcc, "-1",
    SL_COMPARE_V("v16", "s[72:73]")
    "1:\n"
    SL_SELECT("s[72:73]")
    "s_sub_u32 m0, m0, 1\n s_cbranch_scc1 8f\n"
    SL_COMPARE_V("v17", "s[74:75]")
    SL_FINISH("v16", "1", "s65", "s64")
    SL_SELECT("s[74:75]")
    "s_sub_u32 m0, m0, 1\n s_cbranch_scc1 8f\n"
    SL_COMPARE_V("v16", "s[72:73]")
    SL_FINISH("v17", "1", "s64", "s65")
    "s_branch 1b\n8:\n")

// ... 17 lanes only (is it the 64-lane fetch?)
KERNEL(k_slot_17, "0x1ffff",
    SL_COMPARE("v16", "s[72:73]")
    "1:\n"
    SL_SELECT("s[72:73]")
    "s_sub_u32 m0, m0, 1\n s_cbranch_scc1 8f\n"
    SL_COMPARE("v17", "s[74:75]")
    SL_FINISH("v16", "1", "s65", "s64")
    SL_SELECT("s[74:75]")
    "s_sub_u32 m0, m0, 1\n s_cbranch_scc1 8f\n"
    SL_COMPARE("v16", "s[72:73]")
    SL_FINISH("v17", "1", "s64", "s65")
    "s_branch 1b\n8:\n")

// ... without the miss test
#define SL_SELECT_NM(vccp) "v_readlane_b32 s67, v13, s63\n s_lshr_b64 s[70:71], " vccp ", s67\n s_ff1_i32_b32 s68, s70\n s_add_u32 s67, s67, s68\n" TAKE_S
KERNEL(k_slot_nomiss, "-1",
    SL_COMPARE("v16", "s[72:73]")
    "1:\n"
    SL_SELECT_NM("s[72:73]")
    "s_sub_u32 m0, m0, 1\n s_cbranch_scc1 8f\n"
    SL_COMPARE("v17", "s[74:75]")
    SL_FINISH("v16", "1", "s65", "s64")
    SL_SELECT_NM("s[74:75]")
    "s_sub_u32 m0, m0, 1\n s_cbranch_scc1 8f\n"
    SL_COMPARE("v16", "s[72:73]")
    SL_FINISH("v17", "1", "s64", "s65")
    "s_branch 1b\n8:\n")

// ---- the design SURVEY 7 step 6 / VERDICT r4 #1 name: lanes = bit offsets x candidate trees.  One BATCH decodes the literal
// that would start at every bit offset 0 .. 31 of the window under each candidate tree (per lane: the canonical compare as a
// running minimum over the 15 length words, then ONE fetch), then the true chain is WALKED with v_readlane: entry | length <<
// 16 of (tree, offset), context -> tree of the next literal, offset += length.  v40 .. v54: word[L] = limit[L] << 16 | base << 4
// | L of the lane's tree (here: every code 5 bits long).  OFFS lanes per tree; a batch is good for offsets <= LIMIT.
#define OP_PRE                                                                                              \
    "v_mov_b32 v40, 1\n v_mov_b32 v41, 2\n v_mov_b32 v42, 3\n v_mov_b32 v43, 4\n"                          \
    "v_mov_b32 v44, 0x80000005\n v_mov_b32 v45, 0x80000006\n v_mov_b32 v46, 0x80000007\n v_mov_b32 v47, 0x80000008\n" \
    "v_mov_b32 v48, 0x80000009\n v_mov_b32 v49, 0x8000000a\n v_mov_b32 v50, 0x8000000b\n v_mov_b32 v51, 0x8000000c\n" \
    "v_mov_b32 v52, 0x8000000d\n v_mov_b32 v53, 0x8000000e\n v_mov_b32 v54, 0x8000000f\n"
#define OP_BATCH(offmask)                                                                                   \
    "v_and_b32 v9, " offmask ", v20\n"                                                                      \
    "v_lshrrev_b64 v[14:15], v9, s[60:61]\n v_bfrev_b32 v14, v14\n v_lshrrev_b32 v16, 1, v14\n v_or_b32 v16, 0xffff, v16\n v_add_u32 v16, 1, v16\n" \
    "v_sub_u32 v18, v40, v16\n v_sub_u32 v19, v41, v16\n v_min_u32 v17, v18, v19\n"                        \
    "v_sub_u32 v18, v42, v16\n v_sub_u32 v19, v43, v16\n v_min3_u32 v17, v17, v18, v19\n"                  \
    "v_sub_u32 v18, v44, v16\n v_sub_u32 v19, v45, v16\n v_min3_u32 v17, v17, v18, v19\n"                  \
    "v_sub_u32 v18, v46, v16\n v_sub_u32 v19, v47, v16\n v_min3_u32 v17, v17, v18, v19\n"                  \
    "v_sub_u32 v18, v48, v16\n v_sub_u32 v19, v49, v16\n v_min3_u32 v17, v17, v18, v19\n"                  \
    "v_sub_u32 v18, v50, v16\n v_sub_u32 v19, v51, v16\n v_min3_u32 v17, v17, v18, v19\n"                  \
    "v_sub_u32 v18, v52, v16\n v_sub_u32 v19, v53, v16\n v_min3_u32 v17, v17, v18, v19\n"                  \
    "v_sub_u32 v18, v54, v16\n v_min_u32 v17, v17, v18\n"                                                  \
    "v_add_u32 v17, v17, v16\n v_and_b32 v18, 15, v17\n v_bfe_i32 v19, v17, 4, 12\n v_sub_u32 v22, 32, v18\n" \
    "v_lshrrev_b32 v22, v22, v14\n v_add_u32 v22, v22, v19\n v_lshl_add_u32 v22, v22, 1, v12\n v_and_b32 v22, 0x7fe, v22\n" \
    "ds_read_u16 v21, v22\n s_waitcnt lgkmcnt(0)\n v_lshl_or_b32 v21, v18, 16, v21\n s_mov_b32 s70, 0\n"
// one walk step, context modelling (mode 3): s66 = lanes-per-tree * slot of the current literal's tree, s70 = its bit offset
#define OP_STEP_CTX(endlbl)                                                                                 \
    "s_add_u32 s67, s66, s70\n v_readlane_b32 s69, v21, s67\n s_bfe_u32 s65, s69, 0x3000d\n s_lshl3_add_u32 s63, s65, s64\n s_and_b32 s63, s63, 63\n" \
    "v_readlane_b32 s66, v13, s63\n s_bfe_u32 s68, s69, 0x40010\n s_add_u32 s70, s70, s68\n v_writelane_b32 v31, s69, m0\n" \
    "s_mov_b32 s64, s65\n s_sub_u32 m0, m0, 1\n s_cbranch_scc1 8f\n s_cmp_gt_u32 s70, s71\n s_cbranch_scc1 " endlbl "\n"
// one walk step, one tree (no contexts)
#define OP_STEP_ONE(endlbl)                                                                                 \
    "v_readlane_b32 s69, v21, s70\n s_bfe_u32 s68, s69, 0x40010\n s_add_u32 s70, s70, s68\n v_writelane_b32 v31, s69, m0\n" \
    "s_sub_u32 m0, m0, 1\n s_cbranch_scc1 8f\n s_cmp_gt_u32 s70, s71\n s_cbranch_scc1 " endlbl "\n"
#define OP_TAKE "2:\n s_lshr_b64 s[60:61], s[60:61], s70\n s_or_b32 s61, s61, 0x9e370000\n s_sub_u32 s62, s62, s70\n s_cbranch_scc1 9f\n9:\n"
#define OP_KERNEL(NAME, LIMIT, OFFMASK, SLOTSH, STEP)                                                        \
    KERNEL(NAME, "-1", OP_PRE "s_mov_b32 s71, " LIMIT "\n v_lshrrev_b32 v13, " SLOTSH ", v13\n s_mov_b32 s66, 0\n" \
    "1:\n" OP_BATCH(OFFMASK)                                                                                \
    STEP("2f") STEP("2f") STEP("2f") STEP("2f") STEP("2f") STEP("2f") STEP("2f") STEP("2f")                 \
    OP_TAKE "s_branch 1b\n8:\n")
// (v13 = 16 * (lane & 3) from PRE: >> 0 = 16 lanes per tree, 4 trees; << 1 would be 32 lanes per tree -- PRE's v13 is 16 * slot,
// the two-tree kernels mask it to one bit: slot in {0, 1} -> 0 / 32)
OP_KERNEL(k_op_one_17, "17", "31", "8", OP_STEP_ONE)
OP_KERNEL(k_op_one_31, "31", "31", "8", OP_STEP_ONE)
OP_KERNEL(k_op_ctx4_12, "12", "15", "0", OP_STEP_CTX)
__global__ void k_dummy() {}
KERNEL(k_op_ctx2_17, "-1", OP_PRE "s_mov_b32 s71, 17\n v_and_b32 v13, 16, v13\n v_lshlrev_b32 v13, 1, v13\n s_mov_b32 s66, 0\n"
    "1:\n" OP_BATCH("31")
    OP_STEP_CTX("2f") OP_STEP_CTX("2f") OP_STEP_CTX("2f") OP_STEP_CTX("2f") OP_STEP_CTX("2f") OP_STEP_CTX("2f") OP_STEP_CTX("2f") OP_STEP_CTX("2f")
    OP_TAKE "s_branch 1b\n8:\n")
KERNEL(k_op_ctx2_31, "-1", OP_PRE "s_mov_b32 s71, 31\n v_and_b32 v13, 16, v13\n v_lshlrev_b32 v13, 1, v13\n s_mov_b32 s66, 0\n"
    "1:\n" OP_BATCH("31")
    OP_STEP_CTX("2f") OP_STEP_CTX("2f") OP_STEP_CTX("2f") OP_STEP_CTX("2f") OP_STEP_CTX("2f") OP_STEP_CTX("2f") OP_STEP_CTX("2f") OP_STEP_CTX("2f")
    OP_TAKE "s_branch 1b\n8:\n")

// ---- ONE literal tree (no contexts: quality <= 4 encoders, the 1 GiB reader test): today's loop (.Llit1 of brx_hot.S) and the same
// pipelined -- without a context nothing of literal k + 1 depends on literal k's ENTRY, only on its length: the next compare + fetch
// go out right behind the TAKE, the entry is picked up one literal later (v_readlane -> v_writelane, off the chain)
KERNEL(k_one_base, "0x1ffff",
    "1:\n"
    "v_bfrev_b32 v14, s60\n v_lshrrev_b32 v15, v10, v14\n v_cmp_lt_u32 vcc, v15, v11\n v_lshl_add_u32 v15, v15, 1, v12\n ds_read_u16 v16, v15\n"
    "s_ff1_i32_b32 s68, vcc_lo\n"
    TAKE_S
    "s_waitcnt lgkmcnt(0)\n v_readlane_b32 s69, v16, s68\n s_nop 1\n v_mov_b32 v17, s69\n ds_write_b8 v30, v17 offset:2048\n v_add_u32 v30, 1, v30\n"
    "s_sub_u32 s66, s66, 1\n s_cbranch_scc0 1b\n")
#define ONE_CMP(vs) "v_bfrev_b32 v14, s60\n v_lshrrev_b32 v15, v10, v14\n v_cmp_lt_u32 vcc, v15, v11\n v_lshl_add_u32 v15, v15, 1, v12\n ds_read_u16 " vs ", v15\n s_ff1_i32_b32 s68, vcc_lo\n" TAKE_S
#define ONE_FIN(vs, len) "s_waitcnt lgkmcnt(1)\n v_readlane_b32 s69, " vs ", " len "\n"
KERNEL(k_one_pipe, "0x1ffff",
    ONE_CMP("v16") "s_mov_b32 s72, s68\n"
    "1:\n"
    ONE_CMP("v17") ONE_FIN("v16", "s72") "s_mov_b32 s72, s68\n v_writelane_b32 v31, s69, m0\n s_sub_u32 m0, m0, 1\n s_cbranch_scc1 8f\n"
    ONE_CMP("v16") ONE_FIN("v17", "s72") "s_mov_b32 s72, s68\n v_writelane_b32 v31, s69, m0\n s_sub_u32 m0, m0, 1\n s_cbranch_scc0 1b\n8:\n")
// ... the entry's lane written to the ring directly under a one-lane EXEC (no v_readlane / v_writelane at all)
#define ONE_FIN_X(vs, len) "s_waitcnt lgkmcnt(1)\n s_lshl_b32 exec_lo, 1, " len "\n ds_write_b8 v30, " vs " offset:2048\n s_mov_b32 exec_lo, 0x1ffff\n v_add_u32 v30, 1, v30\n"
KERNEL(k_one_pipe_x, "0x1ffff",
    ONE_CMP("v16") "s_mov_b32 s72, s68\n"
    "1:\n"
    ONE_CMP("v17") ONE_FIN_X("v16", "s72") "s_mov_b32 s72, s68\n s_sub_u32 m0, m0, 1\n s_cbranch_scc1 8f\n"
    ONE_CMP("v16") ONE_FIN_X("v17", "s72") "s_mov_b32 s72, s68\n s_sub_u32 m0, m0, 1\n s_cbranch_scc0 1b\n8:\n")

int main() {
    u64 *o;
    hipMalloc(&o, 64);
#define RUN(K, WHAT) { u64 best = ~0ull; for (int r = 0; r < 5; r++) { hipLaunchKernelGGL(K, dim3(1), dim3(64), 0, 0, o); u64 h; hipMemcpy(&h, o, 8, hipMemcpyDeviceToHost); if (h < best) best = h; } \
      printf("%-72s %7.1f cycles per literal\n", WHAT, best / 1024.0); }
    RUN(k_base, "per-literal loop (round 4, sparse build)");
    RUN(k_slot, "four-slot pipelined loop (as in brx_hot.S)");
    RUN(k_slot_dsw, "  ... SGPR counter, ds_write_b8 per literal instead of v_writelane");
    RUN(k_slot_serial, "  ... not pipelined (fetch waited for at once)");
    RUN(k_slot_vcc, "  ... compare into vcc + s_mov_b64");
    RUN(k_slot_17, "  ... EXEC = 17 lanes");
    RUN(k_slot_nomiss, "  ... no miss test");
    RUN(k_one_base, "ONE tree, today's loop (.Llit1)");
    RUN(k_one_pipe, "ONE tree, pipelined: next compare + fetch behind the TAKE, entry picked up a literal later");
    RUN(k_one_pipe_x, "  ... entry stored from its lane under a one-lane EXEC (no readlane / writelane)");
    RUN(k_op_one_17, "offsets x trees: ONE tree, batch good for offsets <= 17 (4 literals of 5 bits)");
    RUN(k_op_one_31, "  ... for offsets <= 31 (7 literals)");
    RUN(k_op_ctx2_17, "offsets x trees: 2 trees x 32 offsets, context walk, offsets <= 17 (4 literals)");
    RUN(k_op_ctx2_31, "  ... offsets <= 31 (7 literals)");
    RUN(k_op_ctx4_12, "offsets x trees: 4 trees x 16 offsets, context walk, offsets <= 12 (3 literals)");
    return 0;
}
