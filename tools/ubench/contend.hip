// contend.hip -- issue-port contention inside one CU on gfx950: W waves of one workgroup (1..16) run the same pattern;
// ticks per instruction seen by wave 0.  Build: hipcc --offload-arch=gfx950 -O2 contend.hip -o contend
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;
#define TIMED(NAME, BODY)                                                                                          \
    __global__ void NAME(u64 *out) {                                                                               \
        u64 t0, t1;                                                                                                \
        u32 v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, s0 = 1, s1 = 2, s2 = 3, s3 = 4;                              \
        __syncthreads();                                                                                           \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)\n .rept 1024\n" BODY "\n .endr\n s_memtime %1\n s_waitcnt lgkmcnt(0)" \
                     : "=s"(t0), "=s"(t1), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) \
                     : : "vcc", "scc", "memory", "s90", "s91");                                                    \
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3; }                \
    }
TIMED(k_s, "s_add_u32 %6, %6, 1")
TIMED(k_v, "v_add_u32 %2, 1, %2")
TIMED(k_sv, "s_add_u32 %6, %6, 1\n v_add_u32 %2, 1, %2")
TIMED(k_ssv, "s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n v_add_u32 %2, 1, %2")
TIMED(k_svv, "s_add_u32 %6, %6, 1\n v_add_u32 %3, 1, %3\n v_add_u32 %2, 1, %2")
TIMED(k_br, "s_cmp_eq_u32 %6, 0\n s_cbranch_scc1 1f\n1:")
TIMED(k_s_br, "s_add_u32 %6, %6, 1\n s_cmp_eq_u32 %6, 0\n s_cbranch_scc1 1f\n1:")
TIMED(k_v16, "s_mov_b64 exec, 0x1ffff\n v_add_u32 %2, 1, %2\n v_add_u32 %2, 1, %2\n v_add_u32 %2, 1, %2\n s_mov_b64 exec, -1")
TIMED(k_nop, "s_nop 0")
TIMED(k_wait, "s_waitcnt lgkmcnt(0)")
int main() {
    u64 *o; hipMalloc(&o, 64);
    int ws[] = {1, 4, 8, 12, 16};
#define RUN(K, N, WHAT) { printf("%-44s", WHAT); for (int w : ws) { u64 best = ~0ull; for (int r = 0; r < 3; r++) { hipLaunchKernelGGL(K, dim3(1), dim3(64 * w), 0, 0, o); u64 h[2]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost); if (h[0] < best) best = h[0]; } \
      printf("  W=%-2d %6.2f", w, best / 1024.0 / N); } printf("   ticks per instruction\n"); }
    RUN(k_s, 1, "s_add dependent");
    RUN(k_v, 1, "v_add dependent");
    RUN(k_sv, 2, "s_add, v_add");
    RUN(k_ssv, 3, "s_add, s_add, v_add");
    RUN(k_svv, 3, "s_add, v_add, v_add");
    RUN(k_br, 2, "s_cmp, s_cbranch not taken");
    RUN(k_s_br, 3, "s_add, s_cmp, s_cbranch not taken");
    RUN(k_v16, 5, "s_mov exec 17 lanes, 3 v_add, s_mov exec");
    RUN(k_nop, 1, "s_nop 0");
    RUN(k_wait, 1, "s_waitcnt lgkmcnt(0)");
    return 0;
}
