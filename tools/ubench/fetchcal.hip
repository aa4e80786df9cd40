// fetchcal.hip -- what FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc) report for known byte counts in the access patterns of the
// decode kernel (MI355X_MICROARCH.md: "other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern").  Each kernel touches a 2 GiB buffer (beyond L2 + Infinity Cache) exactly once:
//   k_byte_per_line   1 B per lane, every lane its own 64-byte line      (a far back-reference of text: one line per reference)
//   k_byte_per_128    1 B per lane, every lane its own 128-byte pair
//   k_byte_adjacent   1 B per lane, the lanes of a wave adjacent          (the bytes of one copy: 64 B per wave)
//   k_dword_adjacent  4 B per lane adjacent                               (input staging: 256 B per wave)
//   k_b128_adjacent   16 B per lane adjacent                              (direct_far_copy / the guide's calibrated case)
//   k_store_b128      16 B per lane stores                                (flushes)
// Run under  rocprofv3 --pmc FETCH_SIZE  and  --pmc WRITE_SIZE  (separate passes); tools/gpu_fetchcal.sh prints the ratios.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// (`magic` is 255 and the buffer holds zeros: the stores never happen, but the compiler cannot drop the loads)
__global__ void k_byte_per_line(const uint8_t *p, uint32_t *sink, size_t stride, uint32_t magic) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t v = p[i * stride];
    if (v == magic) sink[0] = v;
}
__global__ void k_byte_adjacent(const uint8_t *p, uint32_t *sink, uint32_t magic) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t v = p[i];
    if (v == magic) sink[0] = v;
}
__global__ void k_dword_adjacent(const uint32_t *p, uint32_t *sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t v = p[i];
    if (v == 0x12345678u) sink[0] = v;
}
__global__ void k_b128_adjacent(const uint4 *p, uint32_t *sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 v = p[i];
    if (v.x == 0x12345678u && v.w == 1u) sink[0] = v.y;
}
__global__ void k_store_b128(uint4 *p) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    uint8_t *buf;
    uint32_t *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    const size_t lines = bytes / 64;
    hipLaunchKernelGGL(k_byte_per_line, dim3(lines / 256), dim3(256), 0, 0, buf, sink, (size_t)64, 255u);          // 2^25 lanes, 2 GiB of lines
    hipLaunchKernelGGL(k_byte_per_line, dim3(lines / 512), dim3(256), 0, 0, buf, sink, (size_t)128, 255u);         // 2^24 lanes
    hipLaunchKernelGGL(k_byte_adjacent, dim3(bytes / 256), dim3(256), 0, 0, buf, sink, 255u);                      // 2 GiB of bytes
    hipLaunchKernelGGL(k_dword_adjacent, dim3(bytes / 4 / 256), dim3(256), 0, 0, (const uint32_t *)buf, sink);
    hipLaunchKernelGGL(k_b128_adjacent, dim3(bytes / 16 / 256), dim3(256), 0, 0, (const uint4 *)buf, sink);
    hipLaunchKernelGGL(k_store_b128, dim3(bytes / 16 / 256), dim3(256), 0, 0, (uint4 *)buf);
    hipDeviceSynchronize();
    printf("expected: k_byte_per_line(64) touches %zu lines = %zu B of lines, %zu B useful; (128): %zu lanes; adjacent kernels: %zu B each\n",
           lines, lines * 64, lines, lines / 2, bytes);
    return 0;
}
