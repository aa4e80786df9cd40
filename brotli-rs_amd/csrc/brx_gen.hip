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
typedef uint16_t u16;

__constant__ u32 K_INS_BASE[24] = {0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26, 34, 50, 66, 98, 130, 194, 322, 578, 1090, 2114, 6210, 22594};
__constant__ u8 K_INS_EXTRA[24] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24};
__constant__ u32 K_CPY_BASE[24] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 18, 22, 30, 38, 54, 70, 102, 134, 198, 326, 582, 1094, 2118};
__constant__ u8 K_CPY_EXTRA[24] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24};
// insert&copy symbol = 64 * cell + 8 * (insert code & 7) + (copy code & 7), cell by (insert code >> 3, copy code >> 3):
// the cells with an EXPLICIT distance (RFC 7932 section 5)
__constant__ u8 K_CELL[3][3] = {{2, 3, 6}, {4, 5, 8}, {7, 9, 10}};

struct Writer {  // LSB-first bit writer into the stream's output slot (wave-per-stream kernels: every lane keeps the same
                 // state, lane 0's stores are the ones that count -- they all go to the same addresses with the same bytes)
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

// ======================================================================================================================
// The ADAPTIVE generator (BRX_GEN_ADAPTIVE): what a real encoder's streams look like to a decoder.  One WAVEFRONT per stream.
//   * two passes per meta-block: (1) the LZ77 parse, 64 positions per step -- every lane hashes the four bytes at its own
//     position, probes the table and verifies its candidate; the first lane with a match gives the command, the match is
//     extended 64 bytes per step across the lanes -- with the commands kept in a scratch list and the symbol statistics counted
//     in LDS; (2) prefix codes built FROM those statistics (Huffman, limited to 15 bits the way zlib does it) and sent in
//     complex form with zero runs, then the commands written with them;
//   * TWO literal trees chosen through a real context map (context mode UTF8: the previous two bytes through Lut0 / Lut1),
//     two literal block types with a map row each, taking turns every 1200 / 700 literals through real block-switch commands
//     (two-symbol type and count codes): ~80 switches per MiB of text;
//   * last-distance code 0 and the insert&copy symbols that imply it, where the parse repeats a distance.
// Not an encoder to be judged on ratio (greedy parse, one hash probe) -- a source of streams with the shape of BASELINE config 5
// that needs no committed fixture: bench.py --workload gen_c5x1024, tools/gen_fuzz.py.
#define G2_K1 1200u  // literals per block of type 0 (block count code 20: 753 + 9 bits)
#define G2_K2 700u   // ... of type 1 (block count code 19: 497 + 8 bits)
#define G2_MAXSYM 704u

struct G2Cmd { u32 ins, cpy, dist; };  // dist: 0 = no copy (the meta-block ends with the literals); bit 31 = the last distance again

struct G2Lds {
    u32 lit[2][256], icp[704], dst[64];   // pass 1: counts; pass 2: code (bit-reversed) | length << 16
    u8 lut0[256], lut1[256];
    // Huffman scratch (one alphabet at a time)
    u32 s_cnt[G2_MAXSYM]; u16 s_sym[G2_MAXSYM]; // symbols in use, ascending by (count, symbol)
    u32 w_int[G2_MAXSYM]; u16 par_leaf[G2_MAXSYM], par_int[G2_MAXSYM]; u8 dep_int[G2_MAXSYM];
    u8 len[G2_MAXSYM];
    u32 m;
};

__device__ u32 g2_block_type(u32 ord) { return (ord % (G2_K1 + G2_K2)) < G2_K1 ? 0u : 1u; }

// code lengths (<= 15) of one alphabet from its counts: Huffman over the symbols in use, then zlib's overflow repair.
// Wave-cooperative where it is parallel (the sort), lane 0 alone for the two serial passes.  Result in L.len[0..n).
__device__ void g2_build_lengths(G2Lds &L, const u32 *hist, u32 n) {
    const u32 lane = threadIdx.x;
    for (u32 i = lane; i < n; i += 64u) L.len[i] = 0;
    if (lane == 0u) L.m = 0;
    __syncthreads();
    // rank sort of the symbols in use by (count, symbol)
    for (u32 i = lane; i < n; i += 64u) {
        const u32 c = hist[i];
        if (c == 0u) continue;
        u32 r = 0;
        for (u32 j = 0; j < n; j++) {
            const u32 cj = hist[j];
            r += (cj != 0u && (cj < c || (cj == c && j < i))) ? 1u : 0u;
        }
        L.s_cnt[r] = c;
        L.s_sym[r] = (u16)i;
        atomicAdd(&L.m, 1u);
    }
    __syncthreads();
    const u32 m = L.m;
    if (m == 0u) { if (lane == 0u) L.len[0] = 0; return; }  // (unused alphabet: the caller sends a one-symbol code)
    if (m == 1u) return;                                      // one symbol: length 0, a one-symbol code
    if (lane == 0u) {
        // two-queue Huffman: leaves ascending, internal nodes in the order they are made (ascending too)
        u32 li = 0, ii = 0, ni = 0;
        while (ni < m - 1u) {
            u32 w = 0;
            for (u32 k = 0; k < 2u; k++) {
                const bool leaf = li < m && (ii >= ni || L.s_cnt[li] <= L.w_int[ii]);
                if (leaf) { w += L.s_cnt[li]; L.par_leaf[li] = (u16)ni; li++; }
                else { w += L.w_int[ii]; L.par_int[ii] = (u16)ni; ii++; }
            }
            L.w_int[ni++] = w;
        }
        // depths from the root (the last internal node) down; count the leaves per depth, longer than 15 clamped
        u32 bl[17];
        for (u32 k = 0; k <= 16u; k++) bl[k] = 0;
        L.dep_int[m - 2u] = 0;
        for (u32 k = m - 2u; k-- > 0u;) { const u32 d = L.dep_int[L.par_int[k]] + 1u; L.dep_int[k] = (u8)(d > 60u ? 60u : d); }
        for (u32 r = 0; r < m; r++) {
            u32 d = L.dep_int[L.par_leaf[r]] + 1u;
            if (d > 15u) d = 15u;
            bl[d]++;
        }
        // Leaves clamped to 15 bits over-subscribe the code: the Kraft sum in units of 2^-15 is above 2^15.  zlib's repair step --
        // move one leaf down from the deepest level with room (bits -> bits + 1) and hang one 15-bit leaf next to it -- takes
        // exactly one unit off: -2^(14-bits) for the leaf moved, -1 + 2^(14-bits) for the one re-hung.  As many steps as units.
        // (Until round 4 the steps were counted zlib's way, "overflow -= 2", but over the LEAVES only -- zlib counts internal nodes
        // too -- so codes of meta-blocks of 512 KiB and more, where depths beyond 15 appear, stayed over-subscribed: invalid streams.)
        u32 kraft = 0;
        for (u32 d = 1; d <= 15u; d++) kraft += bl[d] << (15u - d);
        while (kraft > 32768u) {
            u32 bits = 14u;
            while (bl[bits] == 0u) bits--;
            bl[bits]--;
            bl[bits + 1u] += 2u;
            bl[15]--;
            kraft--;
        }
        // the rarest symbols get the longest codes
        u32 r = 0;
        for (u32 d = 15u; d >= 1u; d--)
            for (u32 k = 0; k < bl[d]; k++) L.len[L.s_sym[r++]] = (u8)d;
    }
    __syncthreads();
}

// canonical codes (bit-reversed for the LSB-first writer) from L.len[0..n) into tab[sym] = code | length << 16
__device__ void g2_assign_codes(G2Lds &L, u32 *tab, u32 n) {
    if (threadIdx.x == 0u) {
        u32 bl[16], next[16];
        for (u32 k = 0; k < 16u; k++) bl[k] = 0;
        for (u32 i = 0; i < n; i++) if (L.len[i]) bl[L.len[i]]++;
        u32 code = 0;
        bl[0] = 0;
        for (u32 k = 1; k < 16u; k++) { code = (code + bl[k - 1u]) << 1; next[k] = code; }
        for (u32 i = 0; i < n; i++) {
            const u32 l = L.len[i];
            tab[i] = l ? (rev_bits(next[l]++, l) | (l << 16)) : 0u;
        }
    }
    __syncthreads();
}

// one prefix code on the wire (RFC 7932 section 3.4 / 3.5): a one-symbol simple code, or the complex form -- the code-length
// code is static (lengths 4 for symbols 0..11, 16, 17 and 5 for 12..15: complete), zero runs as chained symbol-17 runs
__device__ void g2_put_zero_run(Writer &w, u32 z) {
    // symbol 17: canonical code 13 of length 4 (the 14 length-4 symbols in order 0..11, 16, 17)
    if (z < 3u) { for (u32 k = 0; k < z; k++) w.put(rev_bits(0, 4), 4); return; }
    u32 digits[5], nd = 0;
    while (z > 10u) { digits[nd++] = (z - 3u) & 7u; z = ((z - 3u) >> 3) + 2u; }
    digits[nd++] = z - 3u;
    while (nd--) { w.put(rev_bits(13, 4), 4); w.put(digits[nd], 3); }
}
__device__ void g2_put_code(Writer &w, const G2Lds &L, u32 n, u32 used, u32 one_sym) {
    if (used <= 1u) {  // simple code, NSYM = 1
        w.put(1, 2);
        w.put(0, 2);
        w.put(used ? one_sym : 0u, 32u - (u32)__clz(n - 1u));
        return;
    }
    w.put(0, 2);  // complex, HSKIP = 0
    // code-length code lengths in the order 1,2,3,4,0,5,17,6,16,7,8,...,15: value 4 = bits "01" (LSB first: 1), value 5 = "1111"
    for (u32 k = 0; k < 18u; k++) {
        const u32 sym = k < 4u ? k + 1u : k == 4u ? 0u : k == 5u ? 5u : k == 6u ? 17u : k == 7u ? 6u : k == 8u ? 16u : k - 2u;
        if (sym >= 12u && sym <= 15u) w.put(15, 4); else w.put(1, 2);
    }
    u32 last = 0;
    for (u32 i = 0; i < n; i++) if (L.len[i]) last = i;
    u32 zeros = 0;
    for (u32 i = 0; i <= last; i++) {
        const u32 l = L.len[i];
        if (l == 0u) { zeros++; continue; }
        g2_put_zero_run(w, zeros);
        zeros = 0;
        if (l < 12u) w.put(rev_bits(l, 4), 4); else w.put(rev_bits(28u + (l - 12u), 5), 5);
    }
}

__device__ u32 g2_load4(const u8 *p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }

// insert&copy symbol of a command; implicit = the cells whose distance is "the last one again" without a distance symbol
__device__ u32 g2_icp_symbol(u32 ins, u32 cpy, bool last_again, u32 &ic, u32 &cc, bool &implicit) {
    ic = code_of(K_INS_BASE, ins);
    cc = code_of(K_CPY_BASE, cpy);
    implicit = last_again && ic < 8u && cc < 16u;
    const u32 cell = implicit ? (cc >> 3) : K_CELL[ic >> 3][cc >> 3];
    return 64u * cell + 8u * (ic & 7u) + (cc & 7u);
}
__device__ u32 g2_dist_symbol(u32 dist, u32 &nb, u32 &extra) {
    const u32 v = dist - 1u + 4u;
    nb = 30u - (u32)__clz(v);
    extra = v & ((1u << nb) - 1u);
    return 16u + 2u * (nb - 1u) + ((v >> nb) & 1u);
}

__global__ __launch_bounds__(64) void brx_generate_adaptive_kernel(BrxGenArgs a, G2Cmd *cmds_all, u32 cmd_cap, const u8 *lut) {
    __shared__ G2Lds L;
    const u32 i = blockIdx.x, lane = threadIdx.x;
    if (i >= a.n) return;
    const u8 *src = a.src + a.src_off[i];
    const u64 len = a.src_off[i + 1] - a.src_off[i];
    Writer w;
    w.out = a.out + a.out_off[i];
    w.cap = a.out_off[i + 1] - a.out_off[i];
    w.n = 0; w.acc = 0; w.nacc = 0; w.overflow = false;
    u32 *table = a.hash + (u64)i * BRX_GEN_HASH;
    G2Cmd *cmds = cmds_all + (u64)i * cmd_cap;
    for (u32 k = lane; k < BRX_GEN_HASH; k += 64u) table[k] = 0xffffffffu;
    for (u32 k = lane; k < 256u; k += 64u) { L.lut0[k] = lut[k]; L.lut1[k] = lut[256u + k]; }
    __syncthreads();
    w.put(1u | ((22u - 17u) << 1), 4);  // WBITS = 22
    if (len == 0) w.put(3, 2);          // ISLAST = 1, ISLASTEMPTY = 1
    const u64 max_dist = (1ull << 22) - 16ull;
    u32 last_dist = 4;                  // the ring starts as 4, 11, 15, 16 (RFC 7932 section 4)
    for (u64 pos = 0; pos < len;) {
        const u64 mlen = len - pos < a.mb_bytes ? len - pos : a.mb_bytes, end = pos + mlen;
        const bool last = end == len;
        // ---------------------------------------------------------------- pass 1: parse + statistics
        for (u32 k = lane; k < 512u; k += 64u) (&L.lit[0][0])[k] = 0;
        for (u32 k = lane; k < 704u; k += 64u) L.icp[k] = 0;
        L.dst[lane] = 0;
        __syncthreads();
        u32 nc = 0, nlit = 0;
        u64 p = pos, lit0 = pos;
        u32 ld = last_dist;
        while (lit0 < end) {
            u64 mpos = end;     // start of the next match (end: none, the rest is literals)
            u32 mlen_ = 0, mdist = 0;
            while (p < end) {
                const u64 q = p + lane;
                const bool can = q + 4u <= end;
                const u32 x = can ? g2_load4(src + q) : 0u;
                const u32 h = (x * 2654435761u) >> 21;
                const u32 cand = can ? table[h] : 0xffffffffu;
                const bool ok = can && cand != 0xffffffffu && (u64)cand < q && q - cand <= max_dist && g2_load4(src + cand) == x;
                const u64 mm = __ballot(ok);
                const u32 first = mm ? (u32)__builtin_ctzll(mm) : 64u;
                if (can && lane <= first) table[h] = (u32)q;  // positions up to (and including) the match start enter the table
                if (mm == 0ull) { p += 64u; continue; }
                mpos = p + first;
                const u32 c0 = (u32)__shfl((int)cand, (int)first);
                mdist = (u32)(mpos - c0);
                // extend: 64 bytes per step
                const u64 lim = end - mpos < 4096u ? end - mpos : 4096u;
                u32 ml = 4;
                while (ml < lim) {
                    const u64 k = ml + lane;
                    const bool same = k < lim && src[c0 + k] == src[mpos + k];
                    const u64 ne = ~__ballot(same);
                    if (ne) { ml += (u32)__builtin_ctzll(ne); break; }
                    ml += 64u;
                }
                mlen_ = ml < lim ? ml : (u32)lim;
                break;
            }
            if (mpos > end) mpos = end;
            // the command: literals [lit0, mpos), then the copy (or none at the end of the meta-block)
            const u32 ins = (u32)(mpos - lit0);
            const bool again = mlen_ != 0u && mdist == ld;
            u32 ic, cc; bool implicit;
            const u32 sym = g2_icp_symbol(ins, mlen_ ? mlen_ : 2u, again, ic, cc, implicit);
            if (lane == 0u) {
                cmds[nc].ins = ins; cmds[nc].cpy = mlen_; cmds[nc].dist = mlen_ ? (again ? 0x80000000u | mdist : mdist) : 0u;
                L.icp[sym]++;
                if (mlen_ && !implicit) {
                    u32 nb, ex;
                    L.dst[again ? 0u : g2_dist_symbol(mdist, nb, ex)]++;
                }
            }
            nc++;
            for (u64 q0 = lit0; q0 < mpos; q0 += 64u) {  // the literals' statistics, by the tree their context selects
                const u64 q = q0 + lane;
                if (q < mpos) {
                    const u32 p1 = q >= 1u ? src[q - 1u] : 0u, p2 = q >= 2u ? src[q - 2u] : 0u;
                    const u32 cid = (u32)L.lut0[p1] | (u32)L.lut1[p2];
                    const u32 tree = (cid >> 5) ^ g2_block_type(nlit + (u32)(q - lit0));
                    atomicAdd(&L.lit[tree][src[q]], 1u);
                }
            }
            nlit += ins;
            if (mlen_ && !again) ld = mdist;
            p = mpos + mlen_;
            // (the positions inside the match enter the table too, 64 at a time: better matches later)
            for (u64 q0 = mpos + 1u; q0 < p && q0 + 4u <= end; q0 += 64u) {
                const u64 q = q0 + lane;
                if (q < p && q + 4u <= end) table[(g2_load4(src + q) * 2654435761u) >> 21] = (u32)q;
            }
            lit0 = p;
            if (mlen_ == 0u) break;  // trailing literals: the meta-block is complete
        }
        __syncthreads();
        // ---------------------------------------------------------------- codes
        // meta-block header up to the prefix codes (lane 0 writes; every lane runs the same code on its own Writer copy, only
        // lane 0's stores count -- see Writer::put)
        w.put(last ? 1 : 0, 1);
        if (last) w.put(0, 1);
        const u32 nib = mlen <= (1u << 16) ? 4u : mlen <= (1u << 20) ? 5u : 6u;
        w.put(nib - 4u, 2);
        w.put(mlen - 1, 4u * nib);
        if (!last) w.put(0, 1);
        w.put(1, 1); w.put(0, 3);                         // NBLTYPESL = 2
        w.put(1, 2); w.put(1, 2); w.put(0, 2); w.put(1, 2);   // block type code: simple, symbols {0, 1}, one bit each
        w.put(1, 2); w.put(1, 2); w.put(19, 5); w.put(20, 5); // block count code: simple, symbols {19, 20}, one bit each
        w.put(1, 1); w.put(G2_K1 - 753u, 9);              // first block count: code 20 ("1"), 1200 = 753 + extra
        w.put(0, 1);                                      // NBLTYPESI = 1
        w.put(0, 1);                                      // NBLTYPESD = 1
        w.put(0, 2); w.put(0, 4);                         // NPOSTFIX = 0, NDIRECT = 0
        w.put(2, 2); w.put(2, 2);                         // context mode UTF8 for both literal block types
        w.put(1, 1); w.put(0, 3);                         // NTREESL = 2
        w.put(0, 1);                                      // RLEMAX = 0
        w.put(1, 2); w.put(1, 2); w.put(0, 1); w.put(1, 1);   // context map code: simple, symbols {0, 1}
        for (u32 bt = 0; bt < 2u; bt++)
            for (u32 cid = 0; cid < 64u; cid++) w.put((cid >> 5) ^ bt, 1);
        w.put(0, 1);                                      // no inverse move-to-front
        w.put(0, 1);                                      // NTREESD = 1
        // the four prefix codes: literal tree 0, literal tree 1, insert&copy, distance -- each built, sent, turned into its table
        for (u32 t = 0; t < 4u; t++) {
            u32 *const tab = t == 0u ? L.lit[0] : t == 1u ? L.lit[1] : t == 2u ? L.icp : L.dst;
            const u32 nsym = t < 2u ? 256u : t == 2u ? 704u : 64u;
            g2_build_lengths(L, tab, nsym);
            const u32 used = L.m, one = used ? L.s_sym[0] : 0u;  // (one symbol in use: length 0, a one-symbol code)
            __syncthreads();
            g2_assign_codes(L, tab, nsym);
            g2_put_code(w, L, nsym, used, one);
            __syncthreads();
        }
        // ---------------------------------------------------------------- pass 2: the commands
        u32 lit_left = G2_K1, btype = 0, ord = 0;
        u64 q = pos;
        ld = last_dist;
        for (u32 c = 0; c < nc; c++) {
            const u32 ins = cmds[c].ins, cpy = cmds[c].cpy, dd = cmds[c].dist;
            const bool again = (dd & 0x80000000u) != 0u;
            u32 ic, cc; bool implicit;
            const u32 sym = g2_icp_symbol(ins, cpy ? cpy : 2u, again, ic, cc, implicit);
            w.put(L.icp[sym] & 0xffffu, L.icp[sym] >> 16);
            w.put(ins - K_INS_BASE[ic], K_INS_EXTRA[ic]);
            w.put((cpy ? cpy : 2u) - K_CPY_BASE[cc], K_CPY_EXTRA[cc]);
            for (u32 k = 0; k < ins; k++, q++, ord++) {
                if (lit_left == 0u) {  // block switch: type "the next one" (symbol 1), count code 19 / 20 by the new type
                    btype ^= 1u;
                    w.put(1, 1);
                    if (btype) { w.put(0, 1); w.put(G2_K2 - 497u, 8); lit_left = G2_K2; }
                    else { w.put(1, 1); w.put(G2_K1 - 753u, 9); lit_left = G2_K1; }
                }
                lit_left--;
                const u32 p1 = q >= 1u ? src[q - 1u] : 0u, p2 = q >= 2u ? src[q - 2u] : 0u;
                const u32 tree = ((((u32)L.lut0[p1] | (u32)L.lut1[p2]) >> 5) ^ btype) & 1u;
                const u32 e = L.lit[tree][src[q]];
                w.put(e & 0xffffu, e >> 16);
            }
            if (cpy) {
                if (!implicit) {
                    if (again) {
                        w.put(L.dst[0] & 0xffffu, L.dst[0] >> 16);
                    } else {
                        u32 nb, ex;
                        const u32 ds = g2_dist_symbol(dd, nb, ex);
                        w.put(L.dst[ds] & 0xffffu, L.dst[ds] >> 16);
                        w.put(ex, nb);
                    }
                }
                if (!again) ld = dd;
                q += cpy;
            }
        }
        last_dist = ld;
        pos = end;
        __syncthreads();
    }
    w.finish();
    if (lane == 0u) {
        a.out_len[i] = w.n;
        a.status[i] = w.overflow ? 25 : 0;
    }
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

void brx_launch_generate_adaptive(const void *src, const uint64_t *src_off, uint32_t n, void *out, const uint64_t *out_off,
                                  uint64_t *out_len, int32_t *status, uint32_t mb_bytes, uint32_t *hash, void *cmds, uint32_t cmd_cap,
                                  const void *context_lut, void *hip_stream) {
    BrxGenArgs a;
    a.src = (const u8 *)src; a.src_off = src_off; a.n = n; a.out = (u8 *)out; a.out_off = out_off; a.out_len = out_len;
    a.status = status; a.header = nullptr; a.header_bits = 0; a.mb_bytes = mb_bytes; a.switches = 1; a.hash = hash;
    hipLaunchKernelGGL(brx_generate_adaptive_kernel, dim3(n), dim3(64), 0, (hipStream_t)hip_stream, a, (G2Cmd *)cmds, cmd_cap,
                       (const u8 *)context_lut);
}
