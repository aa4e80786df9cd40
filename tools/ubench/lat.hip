// lat.hip -- single-wave dependent-chain latencies on gfx950 (cycles per link, s_memtime ticks).
// Build: hipcc --offload-arch=gfx950 -O2 lat.hip -o lat   Run on the GPU box: ./lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;
#define N 2048
__global__ void k_lat(int which, u32 *gbuf, u64 *out, u32 stride_words) {
    __shared__ u32 lds[4096];
    const u32 lane = threadIdx.x;
    for (u32 i = lane; i < 4096; i += 64) lds[i] = ((i * 1103515245u + 12345u) >> 4) & 4095u & ~3u; // byte offsets, dword aligned
    __syncthreads();
    u32 v = lane * 4, s = 0, acc = 0;
    u64 t0 = 0, t1 = 0;
    if (which == 0) { // dependent ds_read_b32 chain (uniform address)
        v = 0;
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < N; i++) asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(v));
        t1 = __builtin_amdgcn_s_memtime();
    } else if (which == 1) { // dependent s_load_dword chain (scalar cache)
        u32 off = 0;
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < N; i++) asm volatile("s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+s"(off) : "s"(gbuf));
        t1 = __builtin_amdgcn_s_memtime();
        acc = off;
    } else if (which == 2) { // readfirstlane -> (hazard) -> readlane -> v_mov hop
        u32 tab = (lane * 7 + 3) & 63;
        v = 5;
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < N; i++)
            asm volatile("v_readfirstlane_b32 %1, %0\n s_nop 3\n v_readlane_b32 %1, %2, %1\n s_nop 1\n v_mov_b32 %0, %1" : "+v"(v), "+s"(s) : "v"(tab));
        t1 = __builtin_amdgcn_s_memtime();
    } else if (which == 3) { // dependent VALU chain
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < N; i++) asm volatile("v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0\n v_add_u32 %0, 1, %0" : "+v"(v));
        t1 = __builtin_amdgcn_s_memtime();
    } else if (which == 4) { // dependent SALU chain
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < N; i++) asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1" : "+s"(s));
        t1 = __builtin_amdgcn_s_memtime();
    } else if (which == 5) { // v_cmp -> s_cbranch_vccnz (never taken) + taken s_branch
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < N; i++)
            asm volatile("v_cmp_gt_u32 vcc, 0, %0\n s_cbranch_vccnz 1f\n s_branch 2f\n1: v_add_u32 %0, 1, %0\n2: v_add_u32 %0, 1, %0" : "+v"(v) : : "vcc");
        t1 = __builtin_amdgcn_s_memtime();
    } else if (which == 6) { // dependent global_load_dword chain, stride given (L1/L2/HBM by footprint)
        u32 idx = 0;
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < N; i++) {
            u32 r;
            asm volatile("global_load_dword %0, %1, %2\n s_waitcnt vmcnt(0)" : "=v"(r) : "v"(idx * 4u), "s"(gbuf));
            idx = r;
        }
        t1 = __builtin_amdgcn_s_memtime();
        acc = idx;
    } else if (which == 7) { // ds_bpermute chain
        u32 a = lane * 4;
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < N; i++) asm volatile("ds_bpermute_b32 %0, %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(v));
        t1 = __builtin_amdgcn_s_memtime();
        acc = a;
    } else if (which == 8) { // SALU -> VALU -> SALU round trip: v_mov from sgpr, v_add, readfirstlane
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < N; i++) asm volatile("v_mov_b32 %0, %1\n v_add_u32 %0, 1, %0\n s_nop 0\n v_readfirstlane_b32 %1, %0" : "+v"(v), "+s"(s));
        t1 = __builtin_amdgcn_s_memtime();
    } else if (which == 9) { // independent VALU issue rate (8 independent adds)
        u32 a0 = 1, a1 = 2, a2 = 3, a3 = 4;
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < N; i++) asm volatile("v_add_u32 %0, 1, %0\n v_add_u32 %1, 1, %1\n v_add_u32 %2, 1, %2\n v_add_u32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        t1 = __builtin_amdgcn_s_memtime();
        acc = a0 + a1 + a2 + a3;
    } else if (which == 10) { // alternating independent SALU / VALU
        u32 a0 = 1;
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < N; i++) asm volatile("v_add_u32 %0, 1, %0\n s_add_u32 %1, %1, 1\n v_add_u32 %0, 1, %0\n s_add_u32 %1, %1, 1" : "+v"(a0), "+s"(s));
        t1 = __builtin_amdgcn_s_memtime();
        acc = a0;
    } else if (which == 11) { // ds_read_u16 with 64 distinct addresses then readlane select
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < N; i++) asm volatile("ds_read_u16 %0, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xffc, %0" : "+v"(v));
        t1 = __builtin_amdgcn_s_memtime();
    } else if (which == 12) { // buffer_load_ubyte issue + wait, then ds_write_b8: the far-copy shape (stride = far source)
        u32 idx = lane;
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < N; i++) {
            u32 r;
            asm volatile("global_load_ubyte %0, %1, %2\n s_waitcnt vmcnt(0)" : "=v"(r) : "v"(idx), "s"(gbuf));
            idx = (idx + stride_words * 4u + (r & 1u)) & 0x3ffffffu;
        }
        t1 = __builtin_amdgcn_s_memtime();
        acc = idx;
    }
    if (lane == 0) { out[0] = t1 - t0; out[1] = v + s + acc; }
}
int main() {
    const size_t words = 64u << 20; // 256 MiB
    u32 *g; u64 *o;
    hipMalloc(&g, words * 4); hipMalloc(&o, 64);
    const char *names[] = {"ds_read_b32 dependent", "s_load_dword dependent (K$)", "readfirstlane->readlane->v_mov hop", "v_add dependent (x4)",
                           "s_add dependent (x4)", "v_cmp+cbranch(not taken)+s_branch(taken)+v_add", "global_load dependent", "ds_bpermute dependent",
                           "v_mov(s)->v_add->readfirstlane round trip", "4 independent v_add", "alternating v_add/s_add (x2)", "ds_read_u16 64 addrs dependent",
                           "global_load_ubyte 64 lanes stream-like"};
    for (int which = 0; which <= 12; which++) {
        std::vector<u32> strides = {0};
        if (which == 6) strides = {16, 1024, 16384, 262144, 4u << 20};  // footprint = stride * N words
        if (which == 12) strides = {64, 4096, 65536};
        for (u32 st : strides) {
            if (which == 1) { std::vector<u32> h(1024); for (u32 i = 0; i < 1024; i++) h[i] = ((i * 37u + 11u) & 1023u) * 4u; hipMemcpy(g, h.data(), 4096, hipMemcpyHostToDevice); }
            if (which == 6) { std::vector<u32> h(words, 0); u32 idx = 0; for (int i = 0; i < N + 8; i++) { u32 nx = (u32)(((u64)(i + 1) * st) % words); h[idx] = nx; idx = nx; } hipMemcpy(g, h.data(), words * 4, hipMemcpyHostToDevice); }
            u64 best = ~0ull;
            for (int rep = 0; rep < 3; rep++) {
                hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, which, g, o, st);
                u64 h[2]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
                if (h[0] < best) best = h[0];
            }
            printf("%-52s stride %8u words: %8.1f ticks per iteration\n", names[which], st, (double)best / N);
        }
    }
    return 0;
}
