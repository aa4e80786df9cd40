// issue.hip -- per-instruction issue cost of one lone wave on gfx950: blocks of 256 copies of a pattern between two
// s_memtime reads.  Build: hipcc --offload-arch=gfx950 -O2 issue.hip -o issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;
#define TIMED(NAME, BODY)                                                                                          \
    __global__ void NAME(u64 *out, u32 *buf) {                                                                     \
        __shared__ u32 lds[256];                                                                                   \
        lds[threadIdx.x] = threadIdx.x * 4;                                                                        \
        __syncthreads();                                                                                           \
        u64 t0, t1;                                                                                                \
        u32 v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, s0 = 1, s1 = 2, s2 = 3, s3 = 4;                              \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)\n .rept 256\n" BODY "\n .endr\n s_memtime %1\n s_waitcnt lgkmcnt(0)" \
                     : "=s"(t0), "=s"(t1), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) \
                     : "s"(buf) : "vcc", "scc", "memory", "s90", "s91", "s92", "s93", "v10", "v11");                             \
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3; }                \
    }
TIMED(k_vdep, "v_add_u32 %2, 1, %2")
TIMED(k_vind, "v_add_u32 %2, 1, %2\n v_add_u32 %3, 1, %3\n v_add_u32 %4, 1, %4\n v_add_u32 %5, 1, %5")
TIMED(k_sdep, "s_add_u32 %6, %6, 1")
TIMED(k_sind, "s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n s_add_u32 %8, %8, 1\n s_add_u32 %9, %9, 1")
TIMED(k_mix, "s_add_u32 %6, %6, 1\n v_add_u32 %2, 1, %2\n s_add_u32 %7, %7, 1\n v_add_u32 %3, 1, %3")
TIMED(k_cbr_nt, "s_cmp_eq_u32 %6, 0\n s_cbranch_scc1 1f\n1:")
TIMED(k_br_taken, "s_branch 1f\n s_nop 0\n1:")
TIMED(k_cbr_taken, "s_cmp_lg_u32 %6, 0\n s_cbranch_scc1 1f\n s_nop 0\n1:")
TIMED(k_vcmp_br, "v_cmp_gt_u32 vcc, 0, %2\n s_cbranch_vccnz 1f\n1:")
TIMED(k_waitcnt, "s_waitcnt lgkmcnt(0)")
TIMED(k_nop, "s_nop 0")
TIMED(k_rfl_rt, "v_readfirstlane_b32 s90, %2\n s_add_u32 s90, s90, 1\n v_mov_b32 %2, s90")
TIMED(k_readlane, "v_readlane_b32 s90, %2, %6\n s_and_b32 %6, s90, 63")
TIMED(k_lds, "ds_read_b32 %2, %2\n s_waitcnt lgkmcnt(0)\n v_and_b32 %2, 0x3fc, %2")
TIMED(k_lds2, "ds_read_b32 %2, %2\n v_add_u32 %3, 1, %3\n v_add_u32 %4, 1, %4\n v_add_u32 %5, 1, %5\n v_add_u32 %3, 1, %3\n s_waitcnt lgkmcnt(0)\n v_and_b32 %2, 0x3fc, %2")
TIMED(k_smem, "s_load_dword s90, %10, 0\n s_waitcnt lgkmcnt(0)\n s_add_u32 %6, %6, s90")
TIMED(k_ff1, "v_cmp_lt_u32 vcc, %3, %2\n s_ff1_i32_b32 s90, vcc_lo\n v_readlane_b32 s91, %2, s90\n s_sub_u32 s92, 32, s90\n v_lshrrev_b32 %3, s92, %3\n v_add_u32 %3, s91, %3")
TIMED(k_call, "s_getpc_b64 s[90:91]\n s_add_u32 s90, s90, 12\n s_addc_u32 s91, s91, 0\n s_setpc_b64 s[90:91]")
TIMED(k_gpridx, "s_set_gpr_idx_on %6, 1\n v_mov_b32 %2, %3\n v_mov_b32 %4, %5\n s_set_gpr_idx_off")
TIMED(k_gpridx1, "s_set_gpr_idx_on %6, 1\n v_mov_b32 %2, %3\n s_set_gpr_idx_off\n v_add_u32 %4, 1, %4")
TIMED(k_bfm_exec, "s_bfm_b64 exec, 5, 3\n v_add_u32 %2, 1, %2\n s_mov_b64 exec, -1")
TIMED(k_readlane2, "v_readlane_b32 s90, %2, 3\n v_add_u32 %3, s90, %3")
TIMED(k_lshl_b64, "v_lshrrev_b64 v[10:11], %6, v[10:11]")
TIMED(k_ffbl, "v_cmp_lt_u32 vcc, %3, %2\n s_ff1_i32_b32 s90, vcc_lo\n s_add_u32 %6, %6, s90")
TIMED(k_vdep_x16, "s_mov_b64 exec, 0xffff\n v_add_u32 %2, 1, %2\n v_add_u32 %2, 1, %2\n v_add_u32 %2, 1, %2\n v_add_u32 %2, 1, %2\n s_mov_b64 exec, -1")
TIMED(k_vdep_x17, "s_mov_b64 exec, 0x1ffff\n v_add_u32 %2, 1, %2\n v_add_u32 %2, 1, %2\n v_add_u32 %2, 1, %2\n v_add_u32 %2, 1, %2\n s_mov_b64 exec, -1")
TIMED(k_vdep_x1, "s_mov_b64 exec, 1\n v_add_u32 %2, 1, %2\n v_add_u32 %2, 1, %2\n v_add_u32 %2, 1, %2\n v_add_u32 %2, 1, %2\n s_mov_b64 exec, -1")
TIMED(k_vdep_x64, "s_mov_b64 exec, -1\n v_add_u32 %2, 1, %2\n v_add_u32 %2, 1, %2\n v_add_u32 %2, 1, %2\n v_add_u32 %2, 1, %2\n s_mov_b64 exec, -1")
TIMED(k_idx_salu, "v_readlane_b32 s90, %2, %6\n s_set_gpr_idx_on s90, 6\n v_cmp_lt_u32 vcc, %3, %4\n v_lshl_add_u32 %5, %5, 1, %4\n s_set_gpr_idx_off")
TIMED(k_win_s, "s_lshr_b64 s[92:93], s[92:93], %6\n v_bfrev_b32 %2, s92\n v_cmp_lt_u32 vcc, %2, %3\n s_ff1_i32_b32 %6, vcc_lo")
TIMED(k_win_v, "v_lshrrev_b64 v[10:11], %6, v[10:11]\n v_bfrev_b32 %2, v10\n v_cmp_lt_u32 vcc, %2, %3\n s_ff1_i32_b32 %6, vcc_lo")
int main() {
    u64 *o; u32 *b;
    hipMalloc(&o, 64); hipMalloc(&b, 4096); hipMemset(b, 0, 4096);
#define RUN(K, N, WHAT) { u64 best = ~0ull; for (int r = 0; r < 3; r++) { hipLaunchKernelGGL(K, dim3(1), dim3(64), 0, 0, o, b); u64 h[2]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost); if (h[0] < best) best = h[0]; } \
      printf("%-64s %7.2f ticks per pattern, %6.2f per instruction\n", WHAT, best / 256.0, best / 256.0 / N); }
    RUN(k_vdep, 1, "v_add dependent");
    RUN(k_vind, 4, "4 x v_add independent");
    RUN(k_sdep, 1, "s_add dependent");
    RUN(k_sind, 4, "4 x s_add independent");
    RUN(k_mix, 4, "s_add / v_add alternating, independent");
    RUN(k_cbr_nt, 2, "s_cmp + s_cbranch_scc1 not taken");
    RUN(k_br_taken, 1, "s_branch taken (skips one s_nop)");
    RUN(k_cbr_taken, 2, "s_cmp + s_cbranch_scc1 taken");
    RUN(k_vcmp_br, 2, "v_cmp + s_cbranch_vccnz not taken");
    RUN(k_waitcnt, 1, "s_waitcnt lgkmcnt(0), nothing outstanding");
    RUN(k_nop, 1, "s_nop 0");
    RUN(k_rfl_rt, 3, "v_readfirstlane -> s_add -> v_mov round trip");
    RUN(k_readlane, 2, "v_readlane (SALU lane select) -> s_and chain");
    RUN(k_lds, 3, "ds_read_b32 -> waitcnt -> v_and chain");
    RUN(k_lds2, 7, "ds_read_b32 + 4 independent v_add -> waitcnt -> v_and");
    RUN(k_smem, 3, "s_load_dword -> waitcnt -> s_add chain");
    RUN(k_ff1, 6, "lookup core: v_cmp, s_ff1, v_readlane, s_sub, v_lshrrev, v_add");
    RUN(k_call, 4, "s_getpc + s_add + s_addc + s_setpc (jump to next)");
    RUN(k_gpridx, 4, "s_set_gpr_idx_on + 2 v_mov + off");
    RUN(k_gpridx1, 4, "s_set_gpr_idx_on + v_mov + off + v_add");
    RUN(k_bfm_exec, 3, "s_bfm_b64 exec + v_add + s_mov exec,-1");
    RUN(k_readlane2, 2, "v_readlane -> v_add (SGPR operand)");
    RUN(k_lshl_b64, 1, "v_lshrrev_b64 dependent");
    RUN(k_ffbl, 3, "v_cmp -> s_ff1 -> s_add");
    RUN(k_vdep_x64, 6, "exec -1 : s_mov exec + 4 dependent v_add + s_mov exec");
    RUN(k_vdep_x17, 6, "exec 17 lanes: same");
    RUN(k_vdep_x16, 6, "exec 16 lanes: same");
    RUN(k_vdep_x1, 6, "exec 1 lane: same");
    RUN(k_idx_salu, 5, "v_readlane -> s_set_gpr_idx_on -> v_cmp, v_lshl_add indexed -> off");
    RUN(k_win_s, 4, "window in SGPRs: s_lshr_b64 -> v_bfrev -> v_cmp -> s_ff1 chain");
    RUN(k_win_v, 4, "window in VGPRs: v_lshrrev_b64 -> v_bfrev -> v_cmp -> s_ff1 chain");
    return 0;
}
