// brx_gen.hip -- on-device Brotli stream GENERATOR: a minimal encoder, one GPU thread per stream.
// The reference has no encoder (README.md:1); this exists so that batches of valid, compressible streams of any size can
// be made on the device from data already there -- benchmarks without committed fixtures, differential fuzzing at scale --
// and every stream it writes is decoded back by the decoder of this library and by the CPU oracle in the tests.
//
// Format of a generated stream (RFC 7932): WBITS = 22; the input cut into meta-blocks of `mb_bytes`; every meta-block
// compressed with ONE block type per category, NPOSTFIX = NDIRECT = 0, one literal and one distance tree, and three
// STATIC complete prefix codes transmitted in complex form (the constant 829 bits of tables/gen_header.bin,
// tools/make_gen_header.py): literals 8 bits each, insert&copy symbols 9 bits (symbols 0..319) or 10 bits, distance
// symbols 6 bits; optionally (BRX_GEN_SWITCHES) two literal block types that take turns every 100 literals -- the block-switch
// commands of the format with one-symbol type and count codes, 4 bits per switch.  Commands come from a greedy LZ77 parse: a hash table of the last position of every 4-byte sequence
// (2048 entries per stream), the match extended as far as it goes (<= 16 384 bytes, inside the meta-block), everything
// between matches as the literals of the next command; explicit distances only (no ring codes, no dictionary).
// So the compression is LZ77's alone (alice29: 65 % of the input), the entropy codes are flat -- enough to exercise every part of a
// decoder's command loop with real back-reference statistics.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "brx_device.h"

namespace {

typedef uint32_t u32;
typedef uint64_t u64;
typedef uint8_t u8;

__constant__ u32 K_INS_BASE[24] = {0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26, 34, 50, 66, 98, 130, 194, 322, 578, 1090, 2114, 6210, 22594};
__constant__ u8 K_INS_EXTRA[24] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24};
__constant__ u32 K_CPY_BASE[24] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 18, 22, 30, 38, 54, 70, 102, 134, 198, 326, 582, 1094, 2118};
__constant__ u8 K_CPY_EXTRA[24] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24};
// insert&copy symbol = 64 * cell + 8 * (insert code & 7) + (copy code & 7), cell by (insert code >> 3, copy code >> 3):
// the cells with an EXPLICIT distance (RFC 7932 section 5)
__constant__ u8 K_CELL[3][3] = {{2, 3, 6}, {4, 5, 8}, {7, 9, 10}};

struct Writer {  // LSB-first bit writer into the stream's output slot
    u8 *out;
    u64 cap, n;  // capacity in bytes, bytes written
    u64 acc;
    u32 nacc;
    bool overflow;
    __device__ void put(u64 value, u32 nbits) {  // nbits <= 32
        acc |= value << nacc;
        nacc += nbits;
        while (nacc >= 8u) {
            if (n < cap) out[n] = (u8)acc; else overflow = true;
            n++;
            acc >>= 8;
            nacc -= 8u;
        }
    }
    __device__ void finish() { if (nacc) put(0, 8u - nacc); }
};

__device__ u32 rev_bits(u32 v, u32 n) { return __brev(v) >> (32u - n); }

__device__ u32 code_of(const u32 *base, u32 v) {  // the largest code whose base is <= v
    u32 c = 0;
    for (u32 k = 1; k < 24u; k++) c = base[k] <= v ? k : c;
    return c;
}

// one command: `ins` literals starting at src[lit], then a copy of `cpy` bytes from `dist` back (dist = 0: the last command
// of a meta-block, whose copy is never executed -- RFC 7932 section 9.3: the meta-block ends with its literals)
// (lit_left: with block switches on, literals left in the current literal block; a switch = the 4 extra bits of the next count)
__device__ void put_command(Writer &w, const u8 *src, u64 lit, u32 ins, u32 cpy, u32 dist, u32 &lit_left) {
    const u32 ic = code_of(K_INS_BASE, ins), cc = code_of(K_CPY_BASE, cpy);
    const u32 sym = 64u * K_CELL[ic >> 3][cc >> 3] + 8u * (ic & 7u) + (cc & 7u);
    if (sym < 320u) w.put(rev_bits(sym, 9), 9); else w.put(rev_bits(640u + (sym - 320u), 10), 10);
    w.put(ins - K_INS_BASE[ic], K_INS_EXTRA[ic]);
    w.put(cpy - K_CPY_BASE[cc], K_CPY_EXTRA[cc]);
    for (u32 k = 0; k < ins; k++) {
        if (lit_left == 0u) { w.put(3, 4); lit_left = 100u; }  // block type +1 (no bits), block count 97 + 3 (4 extra bits)
        lit_left--;                                            // (without block switches lit_left starts at 2^32 - 1)
        w.put(rev_bits(src[lit + k], 8), 8);
    }
    if (dist) {  // explicit distance, NPOSTFIX = NDIRECT = 0 (RFC 7932 section 4)
        const u32 v = dist - 1u + 4u;
        const u32 nb = 30u - (u32)__clz(v);  // = bit length - 2
        const u32 h = (v >> nb) & 1u;
        w.put(rev_bits(16u + 2u * (nb - 1u) + h, 6), 6);
        w.put(v & ((1u << nb) - 1u), nb);
    }
}

}  // namespace

struct BrxGenArgs {
    const u8 *src;
    const u64 *src_off;
    u32 n;
    u8 *out;
    const u64 *out_off;
    u64 *out_len;
    int32_t *status;
    const u8 *header;   // the constant bits of a compressed meta-block (tables/gen_header.bin without its length word)
    u32 header_bits;
    u32 mb_bytes;       // input bytes per meta-block (1 .. 2^24)
    u32 switches;       // 1: the header is variant B (two literal block types taking turns every 100 literals)
    u32 *hash;          // n x BRX_GEN_HASH entries
};

#define BRX_GEN_HASH 2048u

__global__ void brx_generate_kernel(BrxGenArgs a) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const u8 *src = a.src + a.src_off[i];
    const u64 len = a.src_off[i + 1] - a.src_off[i];
    Writer w;
    w.out = a.out + a.out_off[i];
    w.cap = a.out_off[i + 1] - a.out_off[i];
    w.n = 0; w.acc = 0; w.nacc = 0; w.overflow = false;
    u32 *table = a.hash + (u64)i * BRX_GEN_HASH;
    for (u32 k = 0; k < BRX_GEN_HASH; k++) table[k] = 0xffffffffu;
    w.put(1u | ((22u - 17u) << 1), 4);  // WBITS = 22
    if (len == 0) {
        w.put(3, 2);  // ISLAST = 1, ISLASTEMPTY = 1
    }
    const u64 max_dist = (1ull << 22) - 16ull;
    for (u64 pos = 0; pos < len;) {
        const u64 mlen = len - pos < a.mb_bytes ? len - pos : a.mb_bytes, end = pos + mlen;
        const bool last = end == len;
        w.put(last ? 1 : 0, 1);                   // ISLAST
        if (last) w.put(0, 1);                    // ISLASTEMPTY
        const u32 nib = mlen <= (1u << 16) ? 4u : mlen <= (1u << 20) ? 5u : 6u;
        w.put(nib - 4u, 2);
        w.put(mlen - 1, 4u * nib);
        if (!last) w.put(0, 1);                   // ISUNCOMPRESSED
        for (u32 b = 0; b < a.header_bits; b += 8u) {
            const u32 k = a.header_bits - b < 8u ? a.header_bits - b : 8u;
            w.put(a.header[b >> 3] & ((1u << k) - 1u), k);
        }
        u32 lit_left = a.switches ? 100u : 0xffffffffu;
        u64 p = pos, lit = pos;
        while (p < end) {
            u32 best = 0, dist = 0;
            if (p + 4 <= end) {
                const u32 x = (u32)src[p] | ((u32)src[p + 1] << 8) | ((u32)src[p + 2] << 16) | ((u32)src[p + 3] << 24);
                const u32 h = (x * 2654435761u) >> 21;  // 11 bits
                const u32 cand = table[h];
                table[h] = (u32)p;
                if (cand != 0xffffffffu && cand < p && p - cand <= max_dist) {
                    const u64 lim = end - p < 16384 ? end - p : 16384;
                    u32 m = 0;
                    while (m < lim && src[cand + m] == src[p + m]) m++;
                    if (m >= 4u) { best = m; dist = (u32)(p - cand); }
                }
            }
            if (best) {
                put_command(w, src, lit, (u32)(p - lit), best, dist, lit_left);
                p += best;
                lit = p;
            } else {
                p++;
            }
        }
        if (lit < end) put_command(w, src, lit, (u32)(end - lit), 2, 0, lit_left);  // trailing literals: the copy is never reached
        pos = end;
    }
    w.finish();
    a.out_len[i] = w.n;
    a.status[i] = w.overflow ? 25 : 0;  // 25 = the slot was too small (same code as the decoder's)
}

void brx_launch_generate(const void *src, const uint64_t *src_off, uint32_t n, void *out, const uint64_t *out_off,
                         uint64_t *out_len, int32_t *status, const void *header, uint32_t header_bits, uint32_t mb_bytes,
                         uint32_t switches, uint32_t *hash, void *hip_stream) {
    BrxGenArgs a;
    a.src = (const u8 *)src; a.src_off = src_off; a.n = n; a.out = (u8 *)out; a.out_off = out_off; a.out_len = out_len;
    a.status = status; a.header = (const u8 *)header; a.header_bits = header_bits; a.mb_bytes = mb_bytes; a.switches = switches; a.hash = hash;
    // divergent, serial work per thread: small blocks so that the streams spread over all CUs
    hipLaunchKernelGGL(brx_generate_kernel, dim3((n + 31u) / 32u), dim3(32), 0, (hipStream_t)hip_stream, a);
}
