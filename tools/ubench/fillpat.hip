// fillpat.hip -- how fast can W single-wave workgroups each fill their own slot of B KiB, 1 KiB (16 B per lane) per store
// instruction, the way periodic_fill() of the decode kernel does?  Variants: cache policy of the stores, stores per iteration.
// usage: fillpat [waves] [KiB per wave]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int AUX, int UNROLL> __global__ __launch_bounds__(64, 8) void k_fill(unsigned char *out, unsigned kib, unsigned pre) {
    const unsigned lane = threadIdx.x;
    unsigned char *base = out + (size_t)blockIdx.x * kib * 1024u;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, kib * 1024u, 0x00020000);
    u32x4 q = {lane, blockIdx.x, 3u, 4u};
    // `pre`: a dependent scalar chain in front of the stores (the header phase of a stream), in units of ~4 cycles
    unsigned x = blockIdx.x;
    for (unsigned i = 0; i < pre; i++) x = x * 1664525u + 1013904223u;
    q.w = x;
    unsigned off = 16u * lane;
    for (unsigned k = 0; k < kib; k += UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) __builtin_amdgcn_raw_buffer_store_b128(q, r, off + 1024u * u, 0, AUX);
        off += 1024u * UNROLL;
    }
}
// the memset-like pattern: a moving front over the whole buffer
__global__ __launch_bounds__(256) void k_front(u32x4 *out, size_t n16) {
    u32x4 q = {1u, 2u, 3u, 4u};
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256u) out[i] = q;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int AUX, int UNROLL> float run(unsigned char *buf, unsigned waves, unsigned kib, unsigned pre) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_fill<AUX, UNROLL>), dim3(waves), dim3(64), 0, 0, buf, kib, pre);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (rep && ms < best) best = ms;
    }
    return best;
}
int main(int argc, char **argv) {
    const unsigned waves = argc > 1 ? atoi(argv[1]) : 8192, kib = argc > 2 ? atoi(argv[2]) : 172;
    const size_t bytes = (size_t)waves * kib * 1024u;
    unsigned char *buf; CK(hipMalloc(&buf, bytes));
    printf("%u waves x %u KiB = %.1f MB\n", waves, kib, bytes / 1e6);
    float ms;
#define SHOW(name, call) ms = call; printf("%-44s %.4f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6)
    SHOW("plain, 1 store per iteration", (run<0, 1>(buf, waves, kib, 0)));
    SHOW("plain, 4 stores per iteration", (run<0, 4>(buf, waves, kib, 0)));
    SHOW("sc0 (aux 1)", (run<1, 4>(buf, waves, kib, 0)));
    SHOW("nt (aux 2)", (run<2, 4>(buf, waves, kib, 0)));
    SHOW("sc0 nt (aux 3)", (run<3, 4>(buf, waves, kib, 0)));
    SHOW("sc1 (aux 16)", (run<16, 4>(buf, waves, kib, 0)));
    SHOW("sc1 nt (aux 18)", (run<18, 4>(buf, waves, kib, 0)));
    SHOW("sc0 sc1 (aux 17)", (run<17, 4>(buf, waves, kib, 0)));
    SHOW("sc0 sc1 nt (aux 19)", (run<19, 4>(buf, waves, kib, 0)));
    SHOW("plain, 10 K cycles of chain in front", (run<0, 4>(buf, waves, kib, 2500)));
    SHOW("plain, 60 K cycles of chain in front", (run<0, 4>(buf, waves, kib, 15000)));
    SHOW("nt, 60 K cycles of chain in front", (run<2, 4>(buf, waves, kib, 15000)));
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k_front, dim3(256 * 8), dim3(256), 0, 0, (u32x4 *)buf, bytes / 16);
            hipEventRecord(b); hipEventSynchronize(b);
            float t; hipEventElapsedTime(&t, a, b); if (rep && t < best) best = t;
        }
        printf("%-44s %.4f ms  %.0f GB/s\n", "moving front (memset-like), 2048 x 256 threads", best, bytes / best / 1e6);
    }
    return 0;
}
