// twowave.hip -- what the in-workgroup parse / copy split (SURVEY 7 hard part 2, option 2) would pay for its
// hand-overs: two wavefronts of ONE workgroup, a command ring in LDS.  Wave A ("parse") runs a dependent scalar chain of the length
// of a command's parse (the part of brx_hot.S that stays with it), writes a 16-byte command into the ring and publishes the head;
// wave B ("copy") polls the head, does the copy's instructions (address arithmetic, a masked LDS read + write: what .Lcopy /
// LAND_BODY do today) and publishes the tail.  Measured on wave A's clock, cycles per command:
//   solo      one wave does both (today's loop)
//   split     A hands every command to B and never waits (flow control only: ring of 64)
//   split+f   ... and on every 3rd command (a literal run follows a copy: one command in three has literals) A needs the copy LANDED
//             before it may read its two context bytes from the ring: it polls B's tail
// Build: hipcc --offload-arch=gfx950 -O2 twowave.hip -o twowave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;

#define NCMD 2048
// LDS layout (bytes): 0 head, 64 tail, 128.. ring of 64 x 16 B commands, 2048.. 2 KiB "output ring"
// the parse chain: PARSE_N dependent SALU ops with two VALU -> SALU hand-overs and one LDS round trip, ~ a command without its copy
#define PARSE                                                                                                  \
    ".rept 10\n s_add_u32 s60, s60, 1\n s_lshr_b32 s61, s60, 3\n s_and_b32 s61, s61, 63\n .endr\n"              \
    "v_mov_b32 v20, s61\n v_lshlrev_b32 v20, 2, v20\n ds_read_b32 v21, v20 offset:2048\n s_waitcnt lgkmcnt(0)\n" \
    "v_readfirstlane_b32 s62, v21\n s_add_u32 s60, s60, s62\n"                                                 \
    ".rept 10\n s_add_u32 s60, s60, 1\n s_lshr_b32 s61, s60, 3\n s_and_b32 s61, s61, 63\n .endr\n"              \
    "v_readlane_b32 s62, v21, s61\n s_xor_b32 s60, s60, s62\n"
// the copy's share: mask, addresses, ring read, landing store (what stays in wave B)
#define COPYWORK                                                                                               \
    "s_bfm_b64 exec, 9, 3\n s_add_u32 s63, s63, 9\n s_min_u32 s64, s63, 63\n s_cmp_gt_u32 s63, s64\n s_cselect_b32 s63, 0, s63\n" \
    "s_sub_u32 s65, s60, 77\n v_add_u32 v22, s65, v1\n v_and_b32 v22, 2047, v22\n ds_read_u8 v23, v22 offset:2048\n"   \
    "s_mov_b64 exec, -1\n s_add_u32 s66, s66, 9\n s_cmp_ge_u32 s66, s67\n s_cselect_b32 s66, 0, s66\n"           \
    "s_bfm_b64 exec, 9, 3\n v_add_u32 v22, s66, v1\n v_and_b32 v22, 2047, v22\n s_waitcnt lgkmcnt(0)\n ds_write_b8 v22, v23 offset:2048\n s_mov_b64 exec, -1\n"

__global__ void k_solo(u64 *out) {
    __shared__ u32 lds[1024];
    lds[threadIdx.x] = threadIdx.x * 7u;
    __syncthreads();
    u64 t0, t1;
    asm volatile("v_mbcnt_lo_u32_b32 v1, -1, 0\n v_mbcnt_hi_u32_b32 v1, -1, v1\n s_mov_b32 s60, 5\n s_mov_b32 s63, 0\n s_mov_b32 s66, 0\n s_mov_b32 s67, 2000\n"
                 "s_mov_b32 s70, %2\n s_memtime %0\n s_waitcnt lgkmcnt(0)\n"
                 "1:\n" PARSE COPYWORK "s_sub_u32 s70, s70, 1\n s_cbranch_scc0 1b\n"
                 "s_waitcnt lgkmcnt(0)\n s_memtime %1\n s_waitcnt lgkmcnt(0)\n"
                 : "=s"(t0), "=s"(t1) : "n"(NCMD - 1) : "vcc", "scc", "memory", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s70", "v1", "v20", "v21", "v22", "v23");
    if (threadIdx.x == 0) out[0] = t1 - t0;
}

// wave 0 = A, wave 1 = B.  need_every: A waits for B's tail to reach its own count on every need_every-th command (0 = never)
template <int NEED> __global__ void k_split(u64 *out) {
    __shared__ u32 lds[1024];
    for (u32 i = threadIdx.x; i < 1024; i += 128) lds[i] = i * 7u;
    if (threadIdx.x == 0) { lds[0] = 0; lds[16] = 0; }
    __syncthreads();
    const u32 wave = threadIdx.x >> 6;
    u64 t0 = 0, t1 = 0;
    if (wave == 0) {
        asm volatile("v_mbcnt_lo_u32_b32 v1, -1, 0\n v_mbcnt_hi_u32_b32 v1, -1, v1\n s_mov_b32 s60, 5\n s_mov_b32 s71, 0\n s_mov_b32 s72, %3\n"
                     "v_mov_b32 v24, 0\n s_mov_b32 s70, %2\n s_memtime %0\n s_waitcnt lgkmcnt(0)\n"
                     "1:\n" PARSE
                     // flow control: at most 60 commands ahead of B
                     "s_and_b32 s74, s70, 15\n s_cmp_lg_u32 s74, 0\n s_cbranch_scc1 5f\n"   /* (looked at every 16th command only; the ring holds 64) */
                     "2:\n ds_read_b32 v25, v24 offset:64\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 s73, v25\n s_sub_u32 s74, s71, s73\n s_cmp_gt_u32 s74, 40\n s_cbranch_scc1 2b\n5:\n"
                     // the command: 16 bytes into slot head & 63, then the head
                     "s_and_b32 s74, s71, 63\n s_lshl_b32 s74, s74, 4\n v_mov_b32 v26, s74\n v_mov_b32 v28, s60\n v_mov_b32 v29, s61\n v_mov_b32 v30, s62\n v_mov_b32 v31, s71\n"
                     "ds_write_b128 v26, v[28:31] offset:128\n s_add_u32 s71, s71, 1\n v_mov_b32 v27, s71\n ds_write_b32 v24, v27\n"
                     // every NEED-th command: the copy must have LANDED before the literal context is read
                     "s_sub_u32 s72, s72, 1\n s_cbranch_scc0 4f\n s_mov_b32 s72, %3\n"
                     "3:\n ds_read_b32 v25, v24 offset:64\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 s73, v25\n s_cmp_lt_u32 s73, s71\n s_cbranch_scc1 3b\n"
                     "4:\n s_sub_u32 s70, s70, 1\n s_cbranch_scc0 1b\n"
                     "s_waitcnt lgkmcnt(0)\n s_memtime %1\n s_waitcnt lgkmcnt(0)\n"
                     : "=s"(t0), "=s"(t1) : "n"(NCMD - 1), "n"(NEED > 0 ? NEED - 1 : 0x7fffffff)
                     : "vcc", "scc", "memory", "s60", "s61", "s62", "s70", "s71", "s72", "s73", "s74", "v1", "v20", "v21", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
        if ((threadIdx.x & 63) == 0) out[0] = t1 - t0;
    } else {
        asm volatile("v_mbcnt_lo_u32_b32 v1, -1, 0\n v_mbcnt_hi_u32_b32 v1, -1, v1\n s_mov_b32 s60, 5\n s_mov_b32 s63, 0\n s_mov_b32 s66, 0\n s_mov_b32 s67, 2000\n"
                     "v_mov_b32 v24, 0\n s_mov_b32 s71, 0\n"
                     "1:\n ds_read_b32 v25, v24\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 s73, v25\n s_cmp_eq_u32 s73, s71\n s_cbranch_scc1 1b\n"
                     "s_and_b32 s74, s71, 63\n s_lshl_b32 s74, s74, 4\n v_mov_b32 v26, s74\n ds_read_b128 v[28:31], v26 offset:128\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 s60, v28\n"
                     COPYWORK
                     "s_add_u32 s71, s71, 1\n v_mov_b32 v27, s71\n s_waitcnt lgkmcnt(0)\n ds_write_b32 v24, v27 offset:64\n"
                     "s_cmp_lt_u32 s71, %0\n s_cbranch_scc1 1b\n s_waitcnt lgkmcnt(0)\n"
                     : : "n"(NCMD) : "vcc", "scc", "memory", "s60", "s63", "s64", "s65", "s66", "s67", "s71", "s73", "s74", "v1", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
    }
}

int main() {
    u64 *o;
    hipMalloc(&o, 64);
#define RUN(LAUNCH, WHAT) { u64 best = ~0ull; for (int r = 0; r < 5; r++) { LAUNCH; u64 h; hipMemcpy(&h, o, 8, hipMemcpyDeviceToHost); if (h < best) best = h; } \
      printf("%-86s %7.1f cycles per command\n", WHAT, best / (double)NCMD); fflush(stdout); }
    RUN(hipLaunchKernelGGL(k_solo, dim3(1), dim3(64), 0, 0, o), "solo: one wave parses and copies (today)");
    RUN(hipLaunchKernelGGL(k_split<0>, dim3(1), dim3(128), 0, 0, o), "split: wave A parses, hands every command to wave B, never waits for a landing");
    RUN(hipLaunchKernelGGL(k_split<3>, dim3(1), dim3(128), 0, 0, o), "split: ... and waits for the landing on every 3rd command (literals after a copy)");
    RUN(hipLaunchKernelGGL(k_split<1>, dim3(1), dim3(128), 0, 0, o), "split: ... on every command");
    return 0;
}
