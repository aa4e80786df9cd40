// power.hip -- which units the 16 waves of a gfx950 CU share: one 1024-thread workgroup per CU (holding most of the LDS,
// so exactly 16 waves per CU) runs the same instruction pattern for ~10-30 ms; reported: pattern instructions retired
// per CU per cycle at 2.4 GHz (wall time of the kernel), on one CU and on all of them (the same: nothing chip-wide is
// involved, and the clock does not move).  The oldest wave is served first, so a cycle counter inside ONE wave
// (s_memtime) sees no contention at all -- the trap the first reading of this kernel's counters fell into.
// Build: hipcc --offload-arch=gfx950 -O2 power.hip -o power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;
#define KERNEL(NAME, BODY)                                                                                         \
    __global__ void __launch_bounds__(1024) NAME(u64 *out, u32 iters, u32 *buf) {                                            \
        __shared__ u32 lds[24 * 1024];                                                                             \
        lds[threadIdx.x] = threadIdx.x * 4;                                                                        \
        __syncthreads();                                                                                           \
        u64 t0, t1;                                                                                                \
        u32 v0 = (threadIdx.x & 63) * 4, v1 = 1, v2 = 2, v3 = 3, s0 = 1, s1 = 2, s2 = 3, s3 = iters;              \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)\n1:\n .rept 64\n" BODY "\n .endr\n s_sub_u32 %9, %9, 1\n s_cmp_lg_u32 %9, 0\n s_cbranch_scc1 1b\n s_mov_b64 exec, -1\n s_memtime %1\n s_waitcnt lgkmcnt(0)" \
                     : "=s"(t0), "=s"(t1), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) \
                     : "s"(buf) : "vcc", "scc", "memory", "s90", "s91", "s92", "s93", "v10", "v11");                         \
        if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = v0 + v1 + v2 + v3 + s0 + s1 + s2 + lds[5]; } \
        if (threadIdx.x == 960 && blockIdx.x == 0) { out[2] = t1 - t0; out[3] = t1; }  /* the youngest wave of the workgroup */ \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[4] = t1; \
    }
KERNEL(k_s, "s_add_u32 %6, %6, 1")
KERNEL(k_s64, "s_lshr_b64 s[92:93], s[92:93], 1")
KERNEL(k_v64, "v_add_u32 %2, 1, %2")
KERNEL(k_v17, "s_mov_b64 exec, 0x1ffff\n v_add_u32 %2, 1, %2")
KERNEL(k_v1, "s_mov_b64 exec, 1\n v_add_u32 %2, 1, %2")
KERNEL(k_vb64, "v_lshrrev_b64 v[10:11], 1, v[10:11]")
KERNEL(k_nop, "s_nop 0")
KERNEL(k_br, "s_cmp_eq_u32 %6, 0\n s_cbranch_scc1 2f\n2:")
KERNEL(k_rl, "v_readlane_b32 s90, %2, 3")
KERNEL(k_lds16, "s_mov_b64 exec, 0xffff\n ds_read_u16 %3, %2\n s_waitcnt lgkmcnt(0)")
KERNEL(k_lds1, "s_mov_b64 exec, 1\n ds_read_u16 %3, %2\n s_waitcnt lgkmcnt(0)")
KERNEL(k_sleep, "s_sleep 1")
KERNEL(k_s_rl, "s_add_u32 %6, %6, 1\n v_readlane_b32 s90, %2, 3")
KERNEL(k_s_brn, "s_add_u32 %6, %6, 1\n s_cbranch_scc1 2f\n2:")
KERNEL(k_nop_brn, "s_nop 0\n s_cbranch_scc1 2f\n2:")
KERNEL(k_s_brt, "s_add_u32 %6, %6, 1\n s_cbranch_scc0 2f\n s_nop 0\n2:")
KERNEL(k_s_v, "s_add_u32 %6, %6, 1\n v_add_u32 %2, 1, %2")
KERNEL(k_s_vv, "s_add_u32 %6, %6, 1\n v_add_u32 %2, 1, %2\n v_add_u32 %3, 1, %3")
KERNEL(k_s_vcmp, "s_add_u32 %6, %6, 1\n v_cmp_lt_u32 vcc, %2, %3")
KERNEL(k_vcmp, "v_cmp_lt_u32 vcc, %2, %3")
KERNEL(k_vcmp_br, "v_cmp_lt_u32 vcc, %2, %3\n s_cbranch_vccnz 2f\n2:")
KERNEL(k_s_wait, "s_add_u32 %6, %6, 1\n s_waitcnt lgkmcnt(0)")
KERNEL(k_s_smem, "s_add_u32 %6, %6, 1\n s_load_dword s90, %10, 0\n s_waitcnt lgkmcnt(0)")
KERNEL(k_s_lds, "s_add_u32 %6, %6, 1\n ds_read_u16 %3, %2")
KERNEL(k_s_vb64, "s_add_u32 %6, %6, 1\n v_lshrrev_b64 v[10:11], 1, v[10:11]")
KERNEL(k_rfl, "v_readfirstlane_b32 s90, %2")
KERNEL(k_s_wl, "s_add_u32 %6, %6, 1\n v_writelane_b32 %3, s90, 5")
KERNEL(k_s_vs, "s_add_u32 %6, %6, 1\n v_add_u32 %2, %7, %2")
KERNEL(k_ldsw17, "s_mov_b64 exec, 0x1ffff\n ds_write_b8 %4, %3")
KERNEL(k_ldsw1, "s_mov_b64 exec, 1\n ds_write_b8 %4, %3")
KERNEL(k_ldsw64, "s_mov_b64 exec, -1\n ds_write_b8 %4, %3")
KERNEL(k_ldsr17s, "s_mov_b64 exec, 0x1ffff\n ds_read_u8 %3, %4")
KERNEL(k_ldsr17d, "s_mov_b64 exec, 0x1ffff\n ds_read_u16 %3, %2")
KERNEL(k_ldsr17b64, "s_mov_b64 exec, 0x1ffff\n ds_read_b64 v[10:11], %2")
int main() {
    u64 *o; hipMalloc(&o, 64);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUNN(K, IT, N, WHAT) for (int g : {1, cus}) { float ms = 0; u64 h[5]; for (int r = 0; r < 2; r++) { hipEventRecord(e0); hipLaunchKernelGGL(K, dim3(g), dim3(1024), 0, 0, o, IT, (u32 *)o); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, o, 40, hipMemcpyDeviceToHost); } \
      printf("%-44s %3d CU(s) x 16 waves: %7.2f ms = %5.2f instructions per CU per cycle (%d per pattern); s_memtime ticks in the oldest / youngest wave %.2f / %.2f M\n", WHAT, g, ms, (double)(IT) * 64.0 * (N) * 16.0 / (ms * 2.4e6), N, h[0] / 1e6, h[2] / 1e6); }
#define RUN(K, IT, WHAT) RUNN(K, IT, 1, WHAT)
    RUN(k_s, 60000, "s_add chain");
    RUN(k_s64, 60000, "s_lshr_b64 chain");
    RUN(k_v64, 60000, "v_add, 64 lanes");
    RUNN(k_v17, 30000, 2, "s_mov exec + v_add, 17 lanes");
    RUNN(k_v1, 30000, 2, "s_mov exec + v_add, 1 lane");
    RUN(k_vb64, 60000, "v_lshrrev_b64, 64 lanes");
    RUN(k_nop, 60000, "s_nop 0");
    RUNN(k_br, 30000, 2, "s_cmp + s_cbranch not taken");
    RUN(k_rl, 60000, "v_readlane");
    RUNN(k_lds16, 6000, 2, "ds_read_u16 16 lanes + wait");
    RUNN(k_lds1, 6000, 2, "ds_read_u16 1 lane + wait");
    RUN(k_sleep, 10000, "s_sleep 1");
    printf("-- mixes (s_add alone = the scalar ALU's 1 per cycle; s_waitcnt counts as an instruction of the pattern)\n");
    RUNN(k_s_rl, 30000, 2, "s_add + v_readlane");
    RUN(k_rfl, 60000, "v_readfirstlane");
    RUNN(k_s_wl, 30000, 2, "s_add + v_writelane");
    RUNN(k_s_brn, 30000, 2, "s_add + s_cbranch not taken");
    RUNN(k_nop_brn, 30000, 2, "s_nop + s_cbranch not taken");
    RUNN(k_s_brt, 30000, 2, "s_add + s_cbranch taken (+ skipped nop)");
    RUNN(k_s_v, 30000, 2, "s_add + v_add");
    RUNN(k_s_vv, 20000, 3, "s_add + 2 v_add");
    RUNN(k_s_vs, 30000, 2, "s_add + v_add with an SGPR operand");
    RUNN(k_s_vcmp, 30000, 2, "s_add + v_cmp");
    RUN(k_vcmp, 60000, "v_cmp");
    RUNN(k_vcmp_br, 30000, 2, "v_cmp + s_cbranch_vccnz not taken");
    RUNN(k_s_wait, 30000, 2, "s_add + s_waitcnt");
    RUNN(k_s_smem, 10000, 2, "s_add + s_load_dword + wait");
    RUNN(k_s_lds, 30000, 2, "s_add + ds_read_u16 (no wait)");
    RUNN(k_s_vb64, 30000, 2, "s_add + v_lshrrev_b64");
    printf("-- LDS instruction shapes (6000 x 64 each, no waits in between; v2 = 4 * lane, v4 = uniform 2)\n");
    RUNN(k_ldsw64, 6000, 2, "ds_write_b8 one address, 64 lanes");
    RUNN(k_ldsw17, 6000, 2, "ds_write_b8 one address, 17 lanes");
    RUNN(k_ldsw1, 6000, 2, "ds_write_b8 one address, 1 lane");
    RUNN(k_ldsr17s, 6000, 2, "ds_read_u8 one address, 17 lanes");
    RUNN(k_ldsr17d, 6000, 2, "ds_read_u16 distinct dwords, 17 lanes");
    RUNN(k_ldsr17b64, 6000, 2, "ds_read_b64 stride 4 B, 17 lanes");
    return 0;
}
