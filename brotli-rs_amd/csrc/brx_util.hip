// brx_util.hip -- brx_compact_batch: the decoded bytes of a batch without the slack of the capacity slots.
//
// brx_decode_batch leaves stream i at out[out_off[i] .. out_off[i] + out_len[i]) inside a slot of out_off[i+1] - out_off[i]
// bytes (the slot doubles as the stream's window, so it is sized for the worst case).  Whoever ships the results on --
// the ragged RCCL gather of shard.py, a writer of one concatenated file -- wants them back to back.  One
// workgroup per 16 KiB piece of a stream (streams and pieces found by binary search in the destination offsets), 16 bytes
// per lane and step with the DESTINATION 16-byte aligned (the source is read at whatever phase that leaves it; gfx950
// global loads take any alignment), heads and tails by the byte.  HBM-bound: reads len, writes len.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BRX_COMPACT_PIECE 16384u

struct __attribute__((packed, aligned(1))) brx_u128u { uint32_t w[4]; };

// One kernel for the three ragged moves a batch needs (round 6: the node's scatter / gather, brx_node.cpp):
//   item i:  dst[dst_off[i] .. dst_off[i] + len[i])  =  src[src_off[i] .. src_off[i] + len[i])
// The work is cut into 16 KiB pieces of the DENSE coordinate `part_off` = the exclusive prefix sum of len (n entries) with grand
// total `total`:  compaction has dst_off == part_off (slots -> back to back), expansion src_off == part_off (back to back -> slots),
// a permuted gather neither.
__global__ __launch_bounds__(256) void brx_ragged_copy_kernel(const uint8_t *src, const uint64_t *src_off, const uint64_t *len,
                                                              uint8_t *dst, const uint64_t *dst_off, const uint64_t *part_off,
                                                              uint32_t n, uint64_t total) {
    // this workgroup's piece of the dense coordinate
    const uint64_t p0 = (uint64_t)blockIdx.x * BRX_COMPACT_PIECE;
    if (p0 >= total) return;
    const uint64_t p1 = p0 + BRX_COMPACT_PIECE < total ? p0 + BRX_COMPACT_PIECE : total;
    // first item whose dense range ends behind p0: the last i with part_off[i] <= p0 (part_off is non-decreasing)
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1u) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (part_off[mid] <= p0) lo = mid; else hi = mid;
    }
    for (uint32_t i = lo; i < n; i++) {
        const uint64_t d0 = part_off[i];
        if (d0 >= p1) break;
        const uint64_t l = len[i];
        const uint64_t a = d0 > p0 ? d0 : p0, b = d0 + l < p1 ? d0 + l : p1; // dense bytes of item i in this piece
        if (a >= b) continue;
        const uint8_t *s = src + src_off[i] + (a - d0);
        uint8_t *d = dst + dst_off[i] + (a - d0);
        const uint64_t cnt = b - a;
        const uint64_t head = (16u - ((uintptr_t)d & 15u)) & 15u;
        const uint64_t h = head < cnt ? head : cnt;
        if (threadIdx.x < h) d[threadIdx.x] = s[threadIdx.x];
        const uint64_t body = (cnt - h) / 16u;
        for (uint64_t k = threadIdx.x; k < body; k += 256u) {
            const brx_u128u v = *(const brx_u128u *)(s + h + 16u * k);
            *(uint4 *)(d + h + 16u * k) = make_uint4(v.w[0], v.w[1], v.w[2], v.w[3]);
        }
        const uint64_t t0 = h + 16u * body;
        if (t0 + threadIdx.x < cnt) d[t0 + threadIdx.x] = s[t0 + threadIdx.x];
    }
}

void brx_launch_ragged_copy(const void *src, const uint64_t *src_off, const uint64_t *len, void *dst, const uint64_t *dst_off,
                            const uint64_t *part_off, uint32_t n, uint64_t total, void *hip_stream) {
    const uint64_t pieces = (total + BRX_COMPACT_PIECE - 1u) / BRX_COMPACT_PIECE;
    if (pieces == 0) return;
    hipLaunchKernelGGL(brx_ragged_copy_kernel, dim3((unsigned)pieces), dim3(256), 0, (hipStream_t)hip_stream, (const uint8_t *)src, src_off,
                       len, (uint8_t *)dst, dst_off, part_off, n, total);
}

void brx_launch_compact(const void *src, const uint64_t *src_off, const uint64_t *len, void *dst, const uint64_t *dst_off,
                        uint32_t n, uint64_t total, void *hip_stream) {
    brx_launch_ragged_copy(src, src_off, len, dst, dst_off, dst_off, n, total, hip_stream);
}
