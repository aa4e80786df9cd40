// brx_kernels.hip -- gfx950 (CDNA4 / MI355X) Brotli decode kernel.
//
// One 64-lane wavefront decodes one stream at a time (persistent grid, streams handed out through an
// atomic counter).  The wave runs the reference's decode path (ende76/brotli-rs, src/lib.rs:1545-2170)
// as wave-uniform code:
//   * the compressed input is staged 256 B at a time ACROSS THE LANES of two VGPRs (one coalesced load
//     per 2048 bits) and fed to a 64-bit scalar bit window with v_readlane -- no per-symbol memory access
//     on the input side                                   (reference: src/bitreader/mod.rs)
//   * prefix codes are canonical first-code tables in LDS; a symbol is resolved by ONE ballot: lane L
//     compares the bit-reversed 15-bit window against limit[L], the first set bit of the ballot is the
//     code length                                         (reference: src/huffman/, src/huffman/tree/)
//   * the last 4 KiB of output live in an LDS ring (the sliding window of src/ringbuffer/); literals,
//     LZ77 copies and transformed dictionary words are produced into the ring by all 64 lanes and leave
//     for HBM in address-aligned 1 KiB blocks of 16 B per lane; back-references older than the ring are
//     read back from the stream's own HBM output          (reference: copy_literals, src/lib.rs:1483-1542)
//   * static dictionary + 121 transforms                 (reference: src/dictionary, src/transformation)
//
// Everything observable follows the reference, quirks Q1..Q15 of SURVEY.md section 2.3 included.
// There is no MFMA here: the path is integer/byte work bounded by HBM bandwidth and by the serial
// prefix-code chain of each stream.
#include <hip/hip_runtime.h>

#include "brx_device.h"

typedef unsigned char u8;
typedef unsigned short u16;
typedef unsigned int u32;
typedef unsigned long long u64;

#define RMASK (BRX_RING_BYTES - 1u)
#define TM_BYTES (BRX_TM_WORDS * 4u)

// status codes: keep in sync with include/brx.h
enum {
    ST_OK = 0, ST_CODE_LENGTHS_CHECKSUM = 1, ST_EXPECTED_END_OF_STREAM = 2, ST_EXCEEDED_EXPECTED_BYTES = 3,
    ST_INVALID_BLOCK_COUNT_CODE = 4, ST_INVALID_BLOCK_SWITCH = 5, ST_INVALID_DICT_LENGTH = 6,
    ST_INVALID_SYMBOL = 8, ST_INVALID_TRANSFORM_ID = 9, ST_NON_POSITIVE_DISTANCE = 10,
    ST_LESS_THAN_TWO_NONZERO = 11, ST_NO_CODE_LENGTH = 12, ST_NON_ZERO_FILL_BIT = 13,
    ST_NON_ZERO_RESERVED_BIT = 14, ST_NON_ZERO_TRAILER_BIT = 15, ST_NON_ZERO_TRAILER_NIBBLE = 16,
    ST_PARSE_CONTEXT_MAP = 17, ST_PARSE_COMPLEX_LENGTHS = 18, ST_PARSE_DISTANCE_CODE = 19,
    ST_PARSE_IAC = 20, ST_PARSE_LITERALS = 21, ST_RUN_LENGTH_EXCEEDED = 23, ST_EOF = 24,
    ST_OUTPUT_TOO_SMALL = 25, ST_REF_PANIC = 26, ST_WATCHDOG = 27
};

enum { LK_OK = 0, LK_NONE = 1, LK_EOF = 2 };

struct __attribute__((aligned(16))) Lds {
    u8 ring[BRX_RING_BYTES];
    u32 tm[BRX_TM_WORDS];
    u8 lens[BRX_LENS_BYTES];
};

#define FI __device__ __attribute__((always_inline)) inline

// ---- wave-level primitives ---------------------------------------------------------------------------
FI u32 rfl(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
FI u32 rdl(u32 v, u32 lane) { return (u32)__builtin_amdgcn_readlane((int)v, (int)lane); }
FI u64 ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Closed-form code tables (spec section 5 / 6; reference src/lookuptable/mod.rs:59-123, src/lib.rs:962-976),
// packed (base << 5) | extra_bits and held one entry per lane.
__device__ const u32 K_INS[24] = {
    (0u << 5) | 0, (1u << 5) | 0, (2u << 5) | 0, (3u << 5) | 0, (4u << 5) | 0, (5u << 5) | 0, (6u << 5) | 1, (8u << 5) | 1,
    (10u << 5) | 2, (14u << 5) | 2, (18u << 5) | 3, (26u << 5) | 3, (34u << 5) | 4, (50u << 5) | 4, (66u << 5) | 5,
    (98u << 5) | 5, (130u << 5) | 6, (194u << 5) | 7, (322u << 5) | 8, (578u << 5) | 9, (1090u << 5) | 10,
    (2114u << 5) | 12, (6210u << 5) | 14, (22594u << 5) | 24};
__device__ const u32 K_COPY[24] = {
    (2u << 5) | 0, (3u << 5) | 0, (4u << 5) | 0, (5u << 5) | 0, (6u << 5) | 0, (7u << 5) | 0, (8u << 5) | 0, (9u << 5) | 0,
    (10u << 5) | 1, (12u << 5) | 1, (14u << 5) | 2, (18u << 5) | 2, (22u << 5) | 3, (30u << 5) | 3, (38u << 5) | 4,
    (54u << 5) | 4, (70u << 5) | 5, (102u << 5) | 5, (134u << 5) | 6, (198u << 5) | 7, (326u << 5) | 8, (582u << 5) | 9,
    (1094u << 5) | 10, (2118u << 5) | 24};
__device__ const u32 K_BLEN[26] = {
    (1u << 5) | 2, (5u << 5) | 2, (9u << 5) | 2, (13u << 5) | 2, (17u << 5) | 3, (25u << 5) | 3, (33u << 5) | 3,
    (41u << 5) | 3, (49u << 5) | 4, (65u << 5) | 4, (81u << 5) | 4, (97u << 5) | 4, (113u << 5) | 5, (145u << 5) | 5,
    (177u << 5) | 5, (209u << 5) | 5, (241u << 5) | 6, (305u << 5) | 6, (369u << 5) | 7, (497u << 5) | 8,
    (753u << 5) | 9, (1265u << 5) | 10, (2289u << 5) | 11, (4337u << 5) | 12, (8433u << 5) | 13, (16625u << 5) | 24};
// NDBITS / DOFFSET of the static dictionary (spec section 8; reference src/dictionary/mod.rs:1-11)
__device__ const u8 K_NDBITS[25] = {0, 0, 0, 0, 10, 10, 11, 11, 10, 10, 10, 10, 10, 9, 9, 8, 7, 7, 8, 7, 7, 6, 6, 5, 5};
__device__ const u32 K_DOFFSET[25] = {0, 0, 0, 0, 0, 4096, 9216, 21504, 35840, 44032, 53248, 63488, 74752, 87040, 93696,
                                      100864, 104704, 106752, 108928, 113536, 115968, 118528, 119872, 121280, 122016};

// ---- decoder state (wave-uniform unless marked per-lane) -----------------------------------------------
struct Dec {
    u32 lane;
    // input side
    const u32 *in_words; // dword-aligned base (<= first byte of the stream)
    u32 w_end;           // dwords that may be touched
    u64 bitpos, bitend;  // absolute bit indices from in_words
    u32 chunkA, chunkB;  // per-lane: staged dwords [cbase, cbase+64), [cbase+64, cbase+128)
    u32 cbase;
    u64 win;             // dwords ww, ww+1
    u32 ww;
    // output side
    u8 *out;             // stream's output base
    u32 cap;             // capacity (clamped to < 2^32-64)
    u32 pos;             // bytes produced (reference: Decompressor.count_output)
    u32 a;               // (uintptr_t)out & 15: ring/global 16-B alignment skew
    u32 vfl;             // flushed, in skewed coordinates (pos + a)
    u32 window;          // (1<<WBITS)-16
    u32 dist0, dist1, dist2, dist3; // last distances, dist0 most recent
    // table memory
    u32 lds_top, scr_top;
    u32 *scratch;
    // per-lane constant vectors
    u32 v_ins, v_copy, v_blen;
    u32 v_lut0, v_lut1, v_lut2;
    u32 needed; // for ST_OUTPUT_TOO_SMALL
    u64 wd, wd_limit; // loop watchdog: every command / meta-block consumes a bit or emits a byte
    const u8 *t_dict;
    const BrxTransform *t_xforms;
};

// ---- table memory: LDS first, HBM spill beyond ---------------------------------------------------------
FI u32 tm_ld32(const Dec &d, const Lds &s, u32 wa) {
    return wa < BRX_TM_WORDS ? s.tm[wa] : d.scratch[wa - BRX_TM_WORDS];
}
FI void tm_st32(const Dec &d, Lds &s, u32 wa, u32 v) {
    if (wa < BRX_TM_WORDS) s.tm[wa] = v; else d.scratch[wa - BRX_TM_WORDS] = v;
}
FI u32 tm_ld16(const Dec &d, const Lds &s, u32 ha) { // halfword address
    return ha < BRX_TM_WORDS * 2 ? ((const u16 *)s.tm)[ha] : ((const u16 *)d.scratch)[ha - BRX_TM_WORDS * 2];
}
FI void tm_st16(const Dec &d, Lds &s, u32 ha, u32 v) {
    if (ha < BRX_TM_WORDS * 2) ((u16 *)s.tm)[ha] = (u16)v; else ((u16 *)d.scratch)[ha - BRX_TM_WORDS * 2] = (u16)v;
}
FI u32 tm_ld8(const Dec &d, const Lds &s, u32 ba) {
    return ba < TM_BYTES ? ((const u8 *)s.tm)[ba] : ((const u8 *)d.scratch)[ba - TM_BYTES];
}
FI void tm_st8(const Dec &d, Lds &s, u32 ba, u32 v) {
    if (ba < TM_BYTES) ((u8 *)s.tm)[ba] = (u8)v; else ((u8 *)d.scratch)[ba - TM_BYTES] = (u8)v;
}
// Objects never straddle the LDS / HBM boundary.
FI u32 tm_alloc(Dec &d, u32 nwords) {
    if (d.lds_top + nwords <= BRX_TM_WORDS) {
        u32 r = d.lds_top;
        d.lds_top += nwords;
        return r;
    }
    u32 r = BRX_TM_WORDS + d.scr_top;
    d.scr_top += nwords;
    return r;
}

// ---- bit input (reference: src/bitreader/mod.rs:21-303) -------------------------------------------------
FI u32 in_load_chunk(const Dec &d, u32 c) {
    u32 i = c + d.lane;
    return i < d.w_end ? d.in_words[i] : 0u;
}
FI u32 in_word(Dec &d, u32 w) {
    if (w >= d.w_end) return 0u;
    if (w < d.cbase || w >= d.cbase + 256u) { // far seek (metadata skip / uncompressed block)
        d.cbase = w & ~63u;
        d.chunkA = in_load_chunk(d, d.cbase);
        d.chunkB = in_load_chunk(d, d.cbase + 64u);
    }
    while (w >= d.cbase + 128u) {
        d.chunkA = d.chunkB;
        d.cbase += 64u;
        d.chunkB = in_load_chunk(d, d.cbase + 64u);
    }
    u32 k = w - d.cbase;
    return k < 64u ? rdl(d.chunkA, k) : rdl(d.chunkB, k - 64u);
}
FI void in_seek(Dec &d, u64 bitpos) {
    d.bitpos = bitpos;
    d.ww = (u32)(bitpos >> 5);
    u64 lo = in_word(d, d.ww);
    u64 hi = in_word(d, d.ww + 1u);
    d.win = lo | (hi << 32);
}
FI u64 in_remaining(const Dec &d) { return d.bitend - d.bitpos; }
// up to 32 bits at the cursor, NOT masked against the end of the stream
FI u32 in_peek_raw(const Dec &d) { return (u32)(d.win >> (d.bitpos & 31u)); }
FI void in_consume(Dec &d, u32 n) {
    d.bitpos += n;
    if ((u32)(d.bitpos >> 5) != d.ww) {
        d.ww += 1u;
        u64 hi = in_word(d, d.ww + 1u);
        d.win = (d.win >> 32) | (hi << 32);
    }
}
// n <= 24 here (every field of the format); false = fewer than n bits remain (reference: Err(_))
FI bool in_bits(Dec &d, u32 n, u32 &v) {
    if (in_remaining(d) < n) return false;
    v = n ? (in_peek_raw(d) & ((1u << n) - 1u)) : 0u;
    in_consume(d, n);
    return true;
}
// read_u8_from_byte_tail, src/bitreader/mod.rs:257-267
FI u32 in_byte_tail(Dec &d) {
    u32 k = (u32)(d.bitpos & 7u);
    u32 v = 0;
    if (k) in_bits(d, 8u - k, v);
    return v;
}

// ---- prefix codes ----------------------------------------------------------------------------------------
// Table layout in table memory (word address h): 16 header words, then the symbols as u16 in
// (length, symbol) order.  header[0] = kind | max_len << 8 | single_symbol << 16  (kind 0 empty, 1 single,
// 2 general); header[L] (1..15) = limit[L] | base[L] << 16 with
//   limit[L] = (first_code[L] + count[L]) << (15-L)   -- exclusive upper bound of length-L codes, left aligned
//   base[L]  = offset[L] - first_code[L]  (mod 2^16)
// Lookup = reference Tree::lookup_symbol (src/huffman/tree/mod.rs:63-93): zero bits for a single-symbol
// code (Q5); an unassigned codeword of an incomplete code reads max_len+1 bits and yields None (Q15).
FI u32 decode_sym(Dec &d, const Lds &s, u32 h, u32 &sym) {
    u32 hv = tm_ld32(d, s, h + (d.lane & 15u)); // per-lane header word
    u32 h0 = rdl(hv, 0);
    u32 kind = h0 & 3u;
    if (kind == 0u) return LK_NONE;
    if (kind == 1u) {
        sym = h0 >> 16;
        return LK_OK;
    }
    u64 rem = in_remaining(d);
    u32 peek = in_peek_raw(d) & 0x7fffu;
    if (rem < 15u) peek &= (1u << (u32)rem) - 1u;
    u32 v = __brev(peek) >> 17; // first stream bit = MSB of a 15-bit left-aligned code
    u64 m = ballot(v < (hv & 0xffffu)) & 0xfffeull; // lanes 1..15 carry limit[1..15]
    if (m == 0ull) {
        u32 maxlen = (h0 >> 8) & 0xffu;
        return rem >= (u64)(maxlen + 1u) ? LK_NONE : LK_EOF;
    }
    u32 L = (u32)__builtin_ctzll(m);
    if ((u64)L > rem) return LK_EOF;
    u32 base = rdl(hv, L) >> 16;
    u32 idx = ((v >> (15u - L)) + base) & 0xffffu;
    sym = rfl(tm_ld16(d, s, (h + 16u) * 2u + idx));
    in_consume(d, L);
    return LK_OK;
}

// Build a general code from s.lens[0..n) (canonical assignment, reference src/huffman/mod.rs:19-43; the
// bl_count[0] quirk Q7 vanishes under the reference's own masking of the code to `len` bits).
// Precondition (checked by the callers like the reference does): Kraft sum <= 1, at least 2 non-zero lengths.
FI u32 build_code(Dec &d, Lds &s, u32 n) {
    const u32 lane = d.lane;
    u32 cnt = 0; // lane L: number of codes of length L
    for (u32 c = 0; c < n; c += 64u) {
        u32 i = c + lane;
        u32 my = i < n ? s.lens[i] : 0u;
        u64 any = ballot(my != 0u);
        if (any == 0ull) continue;
        for (u32 l = 1; l <= 15u; l++) {
            u64 m = ballot(my == l);
            if (lane == l) cnt += (u32)__builtin_popcountll(m);
        }
    }
    u32 code = 0, off = 0, hv = 0, offv = 0, maxlen = 0, present = 0;
    for (u32 l = 1; l <= 15u; l++) {
        u32 c = rdl(cnt, l);
        u32 limit = (code + c) << (15u - l);
        u32 base = (off - code) & 0xffffu;
        if (lane == l) {
            hv = limit | (base << 16);
            offv = off;
        }
        if (c) {
            maxlen = l;
            present |= 1u << l;
        }
        off += c;
        code = (code + c) << 1;
    }
    u32 nnz = off;
    u32 h = tm_alloc(d, 16u + ((nnz + 1u) >> 1));
    if (lane == 0u) hv = 2u | (maxlen << 8);
    if (lane < 16u) tm_st32(d, s, h + lane, hv);
    const u64 lt = (1ull << lane) - 1ull;
    for (u32 c = 0; c < n; c += 64u) {
        u32 i = c + lane;
        u32 my = i < n ? s.lens[i] : 0u;
        u64 any = ballot(my != 0u);
        if (any == 0ull) continue;
        u32 pr = present;
        while (pr) {
            u32 l = (u32)__builtin_ctz(pr);
            pr &= pr - 1u;
            u64 m = ballot(my == l);
            if (m == 0ull) continue;
            u32 run = rdl(offv, l);
            if (my == l) tm_st16(d, s, (h + 16u) * 2u + run + (u32)__builtin_popcountll(m & lt), i);
            if (lane == l) offv += (u32)__builtin_popcountll(m);
        }
    }
    return h;
}

FI u32 build_single(Dec &d, Lds &s, u32 sym) {
    u32 h = tm_alloc(d, 16u);
    if (d.lane < 16u) tm_st32(d, s, h + d.lane, d.lane == 0u ? (1u | (sym << 16)) : 0u);
    return h;
}

FI void lens_clear(const Dec &d, Lds &s, u32 n) {
    for (u32 i = d.lane * 4u; i < n; i += 256u) *(u32 *)&s.lens[i] = 0u;
}

// parse_complex_prefix_code, src/lib.rs:667-875 (Q6, Q15): code-length code over {0..17}, then the lengths.
// Fills s.lens[0..alphabet); the caller builds the table.
FI u32 read_complex_lens(Dec &d, Lds &s, u32 kind, u32 alphabet) {
    u32 v;
    const u32 hskip = kind;
    u32 cl_lo = 0, cl_hi = 0; // 18 x 3-bit code lengths packed: symbols 0..9 in cl_lo, 10..17 in cl_hi
    u32 sum = 0, nonzero = 0, single = 0;
    // order of transmission: 1,2,3,4,0,5,17,6,16,7,8,...,15 (src/lib.rs:669)
    const u64 ORDER = 0xfedcba987ull;        // symbols 7..15 for positions 9..17
    for (u32 i = hskip; i < 18u; i++) {
        // fixed code (src/lib.rs:120-125), stream order: 00->0 01->3 10->4 110->2 1110->1 1111->5
        u64 rem = in_remaining(d);
        u32 p = in_peek_raw(d) & 15u;
        u32 len, val;
        if ((p & 1u) == 0u) { len = 2; val = (p & 2u) ? 3u : 0u; }
        else if ((p & 2u) == 0u) { len = 2; val = 4u; }
        else if ((p & 4u) == 0u) { len = 3; val = 2u; }
        else { len = 4; val = (p & 8u) ? 5u : 1u; }
        // bits beyond the end may be garbage; the reference would hit EOF while walking: the walk reads
        // exactly `len` bits when they exist.  With fewer real bits than the shortest consistent code it fails.
        if (rem < 4u) {
            // re-derive with only `rem` real bits: a code of length len needs len bits
            u32 pm = p & ((1u << (u32)rem) - 1u);
            u32 need;
            if (rem < 2u) need = 2;
            else if ((pm & 1u) == 0u) need = 2;
            else if ((pm & 2u) == 0u) need = 2;
            else if (rem < 3u) need = 3;
            else if ((pm & 4u) == 0u) need = 3;
            else need = 4;
            if ((u64)need > rem) return ST_EOF;
        }
        in_consume(d, len);
        u32 symi = i < 4u ? i + 1u : i == 4u ? 0u : i == 5u ? 5u : i == 6u ? 17u : i == 7u ? 6u : i == 8u ? 16u
                   : (u32)((ORDER >> (4u * (i - 9u))) & 15u);
        if (val) {
            if (symi < 10u) cl_lo |= val << (3u * symi); else cl_hi |= val << (3u * (symi - 10u));
            sum += 32u >> val;
            nonzero++;
            single = symi;
            if (sum == 32u) break;
            if (sum > 32u) return ST_CODE_LENGTHS_CHECKSUM;
        }
    }
    if (nonzero == 0u) return ST_NO_CODE_LENGTH;
    if (nonzero >= 2u && sum < 32u) return ST_CODE_LENGTHS_CHECKSUM;

    // 5-bit lookup table for the code-length code, one entry per lane (lanes 0..31): (symbol << 4) | len
    u32 cltab = 0;
    if (nonzero >= 2u) {
        u32 cp = 0; // 6 x 5-bit counts per length
        for (u32 sy = 0; sy < 18u; sy++) {
            u32 l = sy < 10u ? (cl_lo >> (3u * sy)) & 7u : (cl_hi >> (3u * (sy - 10u))) & 7u;
            cp += 1u << (5u * l);
        }
        u64 np = 0; // next canonical code per length, 8 bits each
        u32 code = 0;
        for (u32 l = 1; l <= 5u; l++) {
            code = (code + (l == 1u ? 0u : (cp >> (5u * (l - 1u))) & 31u)) << 1;
            np |= (u64)code << (8u * l);
        }
        for (u32 sy = 0; sy < 18u; sy++) {
            u32 l = sy < 10u ? (cl_lo >> (3u * sy)) & 7u : (cl_hi >> (3u * (sy - 10u))) & 7u;
            if (l == 0u) continue;
            u32 cd = (u32)(np >> (8u * l)) & 255u;
            np += 1ull << (8u * l);
            u32 rev = __brev(cd) >> (32u - l);
            if ((d.lane & ((1u << l) - 1u)) == rev) cltab = (sy << 4) | l;
        }
    }

    lens_clear(d, s, alphabet);
    u32 total = 0, i = 0, nz = 0;
    u32 last_symbol = 0xffu, last_repeat = 0, last_nz = 8;
    while (i < alphabet) {
        u32 sym;
        if (nonzero == 1u) {
            sym = single; // single-symbol code: zero bits (Q5)
        } else {
            u64 rem = in_remaining(d);
            u32 e = rdl(cltab, in_peek_raw(d) & 31u);
            u32 l = e & 15u;
            if (rem < 5u) { // garbage-safe: decide with the real bits only
                u32 pm = in_peek_raw(d) & ((1u << (u32)rem) - 1u);
                e = rdl(cltab, pm);
                l = e & 15u;
                if ((u64)l > rem) return ST_EOF;
            }
            sym = e >> 4;
            in_consume(d, l);
        }
        if (sym <= 15u) {
            if (d.lane == 0u) s.lens[i] = (u8)sym;
            i++;
            last_symbol = sym;
            last_repeat = 0;
            if (sym) {
                nz++;
                last_nz = sym;
                total += 32768u >> sym;
                if (total == 32768u) break;
                if (total > 32768u) return ST_CODE_LENGTHS_CHECKSUM;
            }
        } else if (sym == 16u) {
            if (!in_bits(d, 2, v)) return ST_EOF;
            u32 add, new_repeat;
            if (last_symbol == 16u && last_repeat) {
                new_repeat = 4u * (last_repeat - 2u) + v + 3u;
                add = new_repeat - last_repeat;
            } else {
                new_repeat = 3u + v;
                add = new_repeat;
            }
            if (i + add > alphabet) return ST_PARSE_COMPLEX_LENGTHS;
            for (u32 k = d.lane; k < add; k += 64u) s.lens[i + k] = (u8)last_nz;
            i += add;
            nz += add;
            total += add * (32768u >> last_nz);
            if (total == 32768u) break;
            if (total > 32768u) return ST_CODE_LENGTHS_CHECKSUM;
            last_repeat = new_repeat;
            last_symbol = 16u;
        } else {
            if (!in_bits(d, 3, v)) return ST_EOF;
            if (last_symbol == 17u && last_repeat) {
                u32 new_repeat = 8u * (last_repeat - 2u) + v + 3u;
                i += new_repeat - last_repeat;
                last_repeat = new_repeat;
            } else {
                last_repeat = 3u + v;
                i += last_repeat;
            }
            if (i > alphabet) return ST_PARSE_COMPLEX_LENGTHS;
            last_symbol = 17u;
        }
    }
    if (nz < 2u) return ST_LESS_THAN_TWO_NONZERO;
    return ST_OK;
}

// parse_prefix_code, src/lib.rs:877-889 = kind (:589-595) + simple (:597-665, Q8) or complex (:667-875, Q6, Q15)
FI u32 read_prefix_code(Dec &d, Lds &s, u32 alphabet, u32 &h) {
    u32 kind, v;
    if (!in_bits(d, 2, kind)) return ST_EOF;
    if (kind == 1u) { // ---- simple
        u32 bit_width = 32u - (u32)__builtin_clz(alphabet - 1u);
        if (!in_bits(d, 2, v)) return ST_EOF;
        u32 nsym = v + 1u;
        u32 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        if (!in_bits(d, bit_width, s0)) return ST_EOF;
        if (s0 >= alphabet) return ST_INVALID_SYMBOL;
        if (nsym > 1u) {
            if (!in_bits(d, bit_width, s1)) return ST_EOF;
            if (s1 >= alphabet) return ST_INVALID_SYMBOL;
        }
        if (nsym > 2u) {
            if (!in_bits(d, bit_width, s2)) return ST_EOF;
            if (s2 >= alphabet) return ST_INVALID_SYMBOL;
        }
        if (nsym > 3u) {
            if (!in_bits(d, bit_width, s3)) return ST_EOF;
            if (s3 >= alphabet) return ST_INVALID_SYMBOL;
        }
        if (nsym > 1u && s0 == s1) return ST_INVALID_SYMBOL;
        if (nsym > 2u && (s0 == s2 || s1 == s2)) return ST_INVALID_SYMBOL;
        if (nsym > 3u && (s0 == s3 || s1 == s3 || s2 == s3)) return ST_INVALID_SYMBOL;
        u32 tree_select = 0;
        if (nsym == 4u && !in_bits(d, 1, tree_select)) return ST_EOF;
        if (nsym == 1u) {
            h = build_single(d, s, s0);
            return ST_OK;
        }
        // Within one length the reference inserts the symbols in ascending order in every NSYM case, so the
        // canonical code is a function of the per-symbol lengths alone.
        lens_clear(d, s, alphabet);
        u32 l0, l1, l2, l3;
        if (nsym == 2u) { l0 = 1; l1 = 1; l2 = 0; l3 = 0; }
        else if (nsym == 3u) { l0 = 1; l1 = 2; l2 = 2; l3 = 0; }
        else if (!tree_select) { l0 = l1 = l2 = l3 = 2; }
        else { l0 = 1; l1 = 2; l2 = 3; l3 = 3; }
        if (d.lane == 0u) {
            s.lens[s0] = (u8)l0;
            s.lens[s1] = (u8)l1;
            if (nsym > 2u) s.lens[s2] = (u8)l2;
            if (nsym > 3u) s.lens[s3] = (u8)l3;
        }
    } else {
        u32 rc = read_complex_lens(d, s, kind, alphabet);
        if (rc) return rc;
    }
    h = build_code(d, s, alphabet);
    return ST_OK;
}

// ---- output: LDS ring + aligned flush to HBM --------------------------------------------------------------
FI void flush_range(Dec &d, const Lds &s, u32 v0, u32 v1) {
    u8 *gbase = d.out - d.a; // 16-B aligned; skewed coordinate v lives at gbase + v
    for (u32 u = (v0 >> 4) + d.lane; u < ((v1 + 15u) >> 4); u += 64u) {
        u32 lo = u * 16u < v0 ? v0 : u * 16u;
        u32 hi = u * 16u + 16u > v1 ? v1 : u * 16u + 16u;
        if (hi - lo == 16u) {
            uint4 q = *(const uint4 *)&s.ring[(u * 16u) & RMASK];
            *(uint4 *)(gbase + (u64)u * 16u) = q;
        } else {
            for (u32 b = lo; b < hi; b++) gbase[b] = s.ring[b & RMASK];
        }
    }
    d.vfl = v1;
}
FI void maybe_flush(Dec &d, const Lds &s) {
    u32 vpos = d.pos + d.a;
    for (;;) {
        u32 blk_end = (d.vfl & ~(BRX_FLUSH_BLOCK - 1u)) + BRX_FLUSH_BLOCK;
        if (vpos < blk_end + BRX_FLUSH_LAG) break;
        flush_range(d, s, d.vfl, blk_end);
    }
}
FI bool out_room(Dec &d, u32 n) {
    if ((u64)d.pos + n > (u64)d.cap) {
        d.needed = (u64)d.pos + n > 0xffffffffull ? 0xffffffffu : d.pos + n;
        return false;
    }
    return true;
}
// last two bytes of the output = reference literal_buf (src/lib.rs:389,407,1361,1726,2117)
FI void ctx_bytes(const Dec &d, const Lds &s, u32 &p1, u32 &p2) {
    u32 v = d.pos + d.a;
    u32 b1 = s.ring[(v - 1u) & RMASK], b2 = s.ring[(v - 2u) & RMASK];
    p1 = d.pos >= 1u ? rfl(b1) : 0u;
    p2 = d.pos >= 2u ? rfl(b2) : 0u;
}

// Window copy, reference copy_literals src/lib.rs:1491-1505: out[pos+i] = out[pos-dist + (i % dist)].
FI void window_copy(Dec &d, Lds &s, u32 dist, u32 len) {
    u32 done = 0, de = dist;
    while (done < len) {
        u32 n = len - done < 64u ? len - done : 64u;
        u32 off = d.lane;
        if (de < n) off = d.lane % de; // overlapped copy shorter than a chunk: periodic source
        u32 back = de - off; // distance of this lane's source byte from pos
        u32 b = 0;
        if (d.lane < n) {
            u32 sp = d.pos - back;
            b = back <= BRX_RING_BYTES ? (u32)s.ring[(sp + d.a) & RMASK] : (u32)d.out[sp];
            s.ring[(d.pos + d.lane + d.a) & RMASK] = (u8)b;
        }
        d.pos += n;
        done += n;
        maybe_flush(d, s);
        if (de < 64u) de = dist * ((63u + dist) / dist); // period-preserving distance >= 64 once 64 bytes exist
    }
}

// Static dictionary word + transform, reference src/lib.rs:1506-1540 and src/transformation/mod.rs:3-209.
// Returns a status; on ST_OK the word (wl bytes) sits in lanes 0..wl-1 of `wbyte`.
FI u32 dict_word(Dec &d, u32 copy_len, u32 word_id, u32 &wl, u32 &wbyte) {
    u32 nbits = K_NDBITS[copy_len];
    u32 index = word_id & ((1u << nbits) - 1u);
    u32 tid = word_id >> nbits;
    if (tid > 120u) return ST_INVALID_TRANSFORM_ID;
    const u8 *wp = d.t_dict + K_DOFFSET[copy_len] + index * copy_len;
    u32 w = d.lane < copy_len ? (u32)wp[d.lane] : 0u;
    const BrxTransform *x = d.t_xforms + tid;
    u32 plen = x->plen, slen = x->slen, op = x->op;
    u32 from = 0, mlen = copy_len, xm = 0;
    if (op == 1u) { // UppercaseFirst, src/transformation/mod.rs:42-82 (Q3: 0x00 first byte -> panic)
        u32 b0 = rdl(w, 0);
        if (b0 == 0u) return ST_REF_PANIC;
        if (b0 >= 97u && b0 <= 122u) { if (d.lane == 0u) xm = 32u; }
        else if (b0 >= 192u && b0 <= 223u) { if (d.lane == 1u) xm = 32u; }
        else if (b0 >= 224u) { if (d.lane == 2u) xm = 5u; }
    } else if (op == 2u) { // UppercaseAll, :3-40
        u32 i = 0;
        while (i < copy_len) {
            u32 b = rdl(w, i);
            if (b < 192u) {
                if (b >= 97u && b <= 122u && d.lane == i) xm = 32u;
                i += 1u;
            } else if (b < 224u) {
                if (d.lane == i + 1u) xm = 32u;
                i += 2u;
            } else {
                if (d.lane == i + 2u) xm = 5u;
                i += 3u;
            }
        }
    } else if (op >= 3u && op <= 11u) { // OmitFirstN: word[min(N, len-1)..] (Q1)
        u32 N = op - 2u;
        from = N < copy_len - 1u ? N : copy_len - 1u;
        mlen = copy_len - from;
    } else if (op >= 12u) { // OmitLastN: word[..max(N,len)-N]
        u32 N = op - 11u;
        mlen = (copy_len > N ? copy_len : N) - N;
    }
    w ^= xm;
    wl = plen + mlen + slen;
    u32 j = d.lane;
    u32 mid = (u32)__shfl((int)w, (int)((j - plen + from) & 63u));
    u32 b;
    if (j < plen) b = x->prefix[j & 7u];
    else if (j < plen + mlen) b = mid;
    else b = x->suffix[(j - plen - mlen) & 7u];
    wbyte = b;
    return ST_OK;
}

// ---- block categories (reference: MetaBlock btype_x / blen_x, src/lib.rs:152-160) -----------------------
struct Cat {
    u32 nbl, btype, btype_prev, blen; // blen 0xffffffff = None (NBLTYPES == 1, Q12)
    u32 h_types, h_counts;
};

// parse_n_bltypes, src/lib.rs:501-525 (also NTREESL / NTREESD)
FI u32 read_n_bltypes(Dec &d, u32 &n) {
    u32 b, k, e;
    if (!in_bits(d, 1, b)) return ST_EOF;
    if (!b) { n = 1; return ST_OK; }
    if (!in_bits(d, 3, k)) return ST_EOF;
    if (k == 0u) { n = 2; return ST_OK; }
    if (!in_bits(d, k, e)) return ST_EOF;
    n = (1u << k) + 1u + e;
    return ST_OK;
}
// parse_block_count, src/lib.rs:957-987 (Ok(None) -> UnexpectedEOF, :977)
FI u32 read_block_count(Dec &d, const Lds &s, u32 h, u32 &blen) {
    u32 sym, e;
    if (decode_sym(d, s, h, sym) != LK_OK) return ST_EOF;
    if (sym > 25u) return ST_INVALID_BLOCK_COUNT_CODE;
    u32 pk = rdl(d.v_blen, sym);
    if (!in_bits(d, pk & 31u, e)) return ST_EOF;
    blen = (pk >> 5) + e;
    return ST_OK;
}
// one symbol of a category: None / Some(0) -> switch / Some(n) -> n-1
// (src/lib.rs:1182-1197, 1294-1306, 1377-1389 + parse_block_switch_command :1226-1250)
FI u32 cat_tick(Dec &d, const Lds &s, Cat &c, bool &switched) {
    switched = false;
    if (c.blen == 0xffffffffu) return ST_OK;
    if (c.blen != 0u) { c.blen--; return ST_OK; }
    u32 code, cnt;
    u32 lk = decode_sym(d, s, c.h_types, code);
    if (lk == LK_NONE) return ST_INVALID_BLOCK_SWITCH;
    if (lk == LK_EOF) return ST_EOF;
    u32 nt = code == 0u ? c.btype_prev : code == 1u ? (c.btype + 1u) % c.nbl : code - 2u;
    u32 rc = read_block_count(d, s, c.h_counts, cnt);
    if (rc) return rc;
    c.btype_prev = c.btype;
    c.btype = nt;
    c.blen = cnt - 1u;
    switched = true;
    return ST_OK;
}

// parse_context_map, src/lib.rs:1070-1144, second half: the run-length coded map itself (the RLEMAX field and
// the prefix code `h` over rlemax+ntrees symbols are read by the header loop).  Values go to table memory
// bytes [cm, cm+len).
FI u32 read_context_map_body(Dec &d, Lds &s, u32 h, u32 rlemax, u32 cm, u32 len) {
    u32 b, v;
    u32 pushed = 0;
    while (pushed < len) {
        u32 code;
        u32 lk = decode_sym(d, s, h, code);
        if (lk == LK_NONE) return ST_PARSE_CONTEXT_MAP;
        if (lk == LK_EOF) return ST_EOF;
        if (code > 0u && code <= rlemax) {
            if (!in_bits(d, code, v)) return ST_EOF;
            u32 repeat = (1u << code) + v;
            if (pushed + repeat > len) return ST_RUN_LENGTH_EXCEEDED;
            for (u32 k = d.lane; k < repeat; k += 64u) tm_st8(d, s, cm + pushed + k, 0u);
            pushed += repeat;
        } else {
            if (d.lane == 0u) tm_st8(d, s, cm + pushed, code == 0u ? 0u : code - rlemax);
            pushed++;
        }
    }
    if (!in_bits(d, 1, b)) return ST_EOF;
    if (b) { // inverse_move_to_front_transform, src/lib.rs:1164-1177: the 256-entry list lives 4 per lane
        u32 m0 = d.lane, m1 = d.lane + 64u, m2 = d.lane + 128u, m3 = d.lane + 192u; // mtf[lane + 64*k]
        for (u32 k = 0; k < len; k++) {
            u32 idx = rfl(tm_ld8(d, s, cm + k));
            u32 q = idx >> 6, l = idx & 63u;
            u32 value = q == 0u ? rdl(m0, l) : q == 1u ? rdl(m1, l) : q == 2u ? rdl(m2, l) : rdl(m3, l);
            if (d.lane == 0u) tm_st8(d, s, cm + k, value);
            // shift mtf[0..idx) up by one, put value in front
            u32 c0 = rdl(m0, 63), c1 = rdl(m1, 63), c2 = rdl(m2, 63);
            u32 s0 = (u32)__shfl_up((int)m0, 1), s1 = (u32)__shfl_up((int)m1, 1), s2 = (u32)__shfl_up((int)m2, 1),
                s3 = (u32)__shfl_up((int)m3, 1);
            if (d.lane == 0u) { s0 = value; s1 = c0; s2 = c1; s3 = c2; }
            if (d.lane <= idx) m0 = s0;
            if (d.lane + 64u <= idx) m1 = s1;
            if (d.lane + 128u <= idx) m2 = s2;
            if (d.lane + 192u <= idx) m3 = s3;
        }
    }
    return ST_OK;
}

FI u32 lut8(u32 vec, u32 b) { return (rdl(vec, b >> 2) >> ((b & 3u) * 8u)) & 0xffu; }

// One compressed meta-block: header (src/lib.rs:1745-2002) + command loop (src/lib.rs:2003-2141).
FI u32 compressed_meta_block(Dec &d, Lds &s, u32 mlen) {
    Cat L, I, D, cur;
    u32 rc, v;
    d.lds_top = 0;
    d.scr_top = 0;
    // ---- header: ONE loop whose tail reads "the next prefix code"; the head consumes the code read by the
    // previous iteration.  Order of fields = order of the reference's states NBltypesL .. PrefixCodesDistances.
    enum { S_CAT_N, S_CAT_TYPES, S_CAT_COUNTS, S_MISC, S_CM, S_CM_BODY, S_NTD, S_CODES_INIT, S_CODE };
    u32 step = S_CAT_N, c = 0, h = 0;
    u32 npostfix = 0, ndirect = 0, cmode_w = 0, ntl = 1, ntd = 1, cml = 0, cmd = 0, dalpha = 0;
    u32 cm = 0, cm_len = 0, rlemax = 0, which = 0, save_lds = 0, save_scr = 0;
    u32 ht = 0, total = 0, idx = 0;
    L.nbl = I.nbl = D.nbl = 1; L.btype = I.btype = D.btype = 0; L.btype_prev = I.btype_prev = D.btype_prev = 1;
    L.blen = I.blen = D.blen = 0xffffffffu; L.h_types = I.h_types = D.h_types = 0; L.h_counts = I.h_counts = D.h_counts = 0;
    cur = L;
    for (;;) {
        u32 alphabet = 0;
        if (step == S_CAT_N) { // parse_n_bltypes_{l,i,d} :527-546 and what follows each (:1745-1885)
            cur.btype = 0; cur.btype_prev = 1; cur.blen = 0xffffffffu; cur.h_types = 0; cur.h_counts = 0;
            if ((rc = read_n_bltypes(d, cur.nbl))) return rc;
            if (cur.nbl >= 2u) {
                alphabet = cur.nbl + 2u;
                step = S_CAT_TYPES;
            } else {
                if (c == 0u) L = cur; else if (c == 1u) I = cur; else D = cur;
                c++;
                step = c < 3u ? S_CAT_N : S_MISC;
                continue;
            }
        } else if (step == S_CAT_TYPES) {
            cur.h_types = h;
            alphabet = 26u;
            step = S_CAT_COUNTS;
        } else if (step == S_CAT_COUNTS) {
            cur.h_counts = h;
            if ((rc = read_block_count(d, s, h, cur.blen))) return rc; // parse_first_block_count_* :989-1014
            if (c == 0u) L = cur; else if (c == 1u) I = cur; else D = cur;
            c++;
            step = c < 3u ? S_CAT_N : S_MISC;
            continue;
        } else if (step == S_MISC) {
            if (!in_bits(d, 2, v)) return ST_EOF; // parse_n_postfix :548
            npostfix = v;
            if (!in_bits(d, 4, v)) return ST_EOF; // parse_n_direct :555
            ndirect = v << npostfix;
            dalpha = 16u + ndirect + (48u << npostfix);
            cmode_w = tm_alloc(d, (L.nbl + 3u) >> 2); // context modes, 2 bits per literal block type :562
            for (u32 i = 0; i < L.nbl; i++) {
                if (!in_bits(d, 2, v)) return ST_EOF;
                if (d.lane == 0u) tm_st8(d, s, cmode_w * 4u + i, v);
            }
            if ((rc = read_n_bltypes(d, ntl))) return rc; // parse_n_trees_l :575
            cml = tm_alloc(d, 16u * L.nbl) * 4u; // 64 bytes per block type, zero = tree 0
            for (u32 k = d.lane; k < 16u * L.nbl; k += 64u) tm_st32(d, s, (cml >> 2) + k, 0u);
            if (ntl >= 2u) {
                cm = cml; cm_len = 64u * L.nbl; which = 0;
                step = S_CM;
                continue;
            } else {
                step = S_NTD;
                continue;
            }
        } else if (step == S_NTD) {
            if ((rc = read_n_bltypes(d, ntd))) return rc; // parse_n_trees_d :582
            cmd = tm_alloc(d, D.nbl) * 4u; // 4 bytes per block type
            for (u32 k = d.lane; k < D.nbl; k += 64u) tm_st32(d, s, (cmd >> 2) + k, 0u);
            if (ntd >= 2u) {
                cm = cmd; cm_len = 4u * D.nbl; which = 1;
                step = S_CM;
                continue;
            } else {
                step = S_CODES_INIT;
                continue;
            }
        } else if (step == S_CODES_INIT) {
            total = ntl + I.nbl + ntd;
            ht = tm_alloc(d, total); // handle table: literal trees | insert&copy trees | distance trees
            idx = 0;
            step = S_CODE;
            alphabet = 256u; // parse_prefix_codes_literals :1016
        } else if (step == S_CODE) {
            if (d.lane == 0u) tm_st32(d, s, ht + idx, h);
            idx++;
            if (idx == total) break;
            // :1016 literals (256), :1034 insert&copy (704), :1052 distances (16 + NDIRECT + 48<<NPOSTFIX)
            alphabet = idx < ntl ? 256u : idx < ntl + I.nbl ? 704u : dalpha;
        }
        else if (step == S_CM) {
            // parse_context_map :1070-1144: RLEMAX, then a prefix code over rlemax+ntrees symbols, then the map
            u32 b;
            if (!in_bits(d, 1, b)) return ST_EOF;
            rlemax = 0;
            if (b) {
                if (!in_bits(d, 4, v)) return ST_EOF;
                rlemax = v + 1u;
            }
            save_lds = d.lds_top; // the map's code is dead once the map is read
            save_scr = d.scr_top;
            alphabet = rlemax + (which ? ntd : ntl);
            step = S_CM_BODY;
        } else if (step == S_CM_BODY) {
            if ((rc = read_context_map_body(d, s, h, rlemax, cm, cm_len))) return rc;
            d.lds_top = save_lds;
            d.scr_top = save_scr;
            step = which ? S_CODES_INIT : S_NTD;
            continue;
        }
        if ((rc = read_prefix_code(d, s, alphabet, h))) return rc;
    }
    const u32 hl = ht, hi = ht + ntl, hd = ht + ntl + I.nbl;

    u32 mb = 0; // MetaBlock.count_output
    u32 h_iac = rfl(tm_ld32(d, s, hi + I.btype));
    u32 cmode = rfl(tm_ld8(d, s, cmode_w * 4u + L.btype));
    bool sw;
    for (;;) {
        if (++d.wd > d.wd_limit) return ST_WATCHDOG;
        // ---- parse_insert_and_copy_length :1179-1208
        if ((rc = cat_tick(d, s, I, sw))) return rc;
        if (sw) h_iac = rfl(tm_ld32(d, s, hi + I.btype));
        u32 sym;
        u32 lk = decode_sym(d, s, h_iac, sym);
        if (lk == LK_NONE) return ST_PARSE_IAC;
        if (lk == LK_EOF) return ST_EOF;
        const bool implicit_zero = sym < 128u; // :2012-2015
        // ---- decode_insert_and_copy_length :1210-1224 (table = spec section 5)
        u32 cell = sym >> 6;
        // cell -> (insert code offset, copy code offset): 0:(0,0) 1:(0,8) 2:(0,0) 3:(0,8) 4:(8,0) 5:(8,8)
        // 6:(0,16) 7:(16,0) 8:(8,16) 9:(16,8) 10:(16,16); one nibble per cell, in units of 8
        u32 ioff = (u32)((0x22120110000ull >> (4u * cell)) & 15u) * 8u;
        u32 coff = (u32)((0x21202101010ull >> (4u * cell)) & 15u) * 8u;
        u32 pki = rdl(d.v_ins, ioff + ((sym >> 3) & 7u));
        u32 pkc = rdl(d.v_copy, coff + (sym & 7u));
        u32 e;
        if (!in_bits(d, pki & 31u, e)) return ST_EOF;
        u32 insert_len = (pki >> 5) + e;
        if (!in_bits(d, pkc & 31u, e)) return ST_EOF;
        u32 copy_len = (pkc >> 5) + e;
        if (mlen < mb + insert_len) return ST_EXCEEDED_EXPECTED_BYTES; // :2036 (Q4)
        if (!out_room(d, insert_len)) return ST_OUTPUT_TOO_SMALL;

        // ---- parse_insert_literals :1286-1365 (+ InsertLiterals state :2048-2081)
        if (insert_len) {
            u32 p1, p2;
            ctx_bytes(d, s, p1, p2);
            for (u32 k = 0; k < insert_len; k++) {
                if ((rc = cat_tick(d, s, L, sw))) return rc;
                if (sw) cmode = rfl(tm_ld8(d, s, cmode_w * 4u + L.btype));
                u32 cid;
                if (cmode == 0u) cid = p1 & 0x3fu;
                else if (cmode == 1u) cid = p1 >> 2;
                else if (cmode == 2u) cid = lut8(d.v_lut0, p1) | lut8(d.v_lut1, p2);
                else cid = (lut8(d.v_lut2, p1) << 3) | lut8(d.v_lut2, p2);
                u32 ti = rfl(tm_ld8(d, s, cml + L.btype * 64u + cid));
                u32 h = rfl(tm_ld32(d, s, hl + ti));
                u32 lit;
                lk = decode_sym(d, s, h, lit);
                if (lk == LK_NONE) return ST_PARSE_LITERALS;
                if (lk == LK_EOF) return ST_EOF;
                if (d.lane == 0u) s.ring[(d.pos + d.a) & RMASK] = (u8)lit;
                d.pos++;
                p2 = p1;
                p1 = lit;
                if (((d.pos + d.a) & 63u) == 0u) maybe_flush(d, s);
            }
            mb += insert_len;
            maybe_flush(d, s);
        }
        if (mb == mlen) break; // :2069: the copy part of the last command is ignored

        // ---- parse_distance_code :1367-1410
        u32 dcode;
        if (implicit_zero) {
            dcode = 0;
        } else {
            if ((rc = cat_tick(d, s, D, sw))) return rc;
            u32 cid = copy_len >= 5u ? 3u : copy_len - 2u;
            u32 ti = rfl(tm_ld8(d, s, cmd + D.btype * 4u + cid));
            u32 h = rfl(tm_ld32(d, s, hd + ti));
            lk = decode_sym(d, s, h, dcode);
            if (lk == LK_NONE) return ST_PARSE_DISTANCE_CODE;
            if (lk == LK_EOF) return ST_EOF;
        }
        // ---- decode_distance :1412-1481
        u32 distance;
        if (dcode <= 3u) {
            distance = dcode == 0u ? d.dist0 : dcode == 1u ? d.dist1 : dcode == 2u ? d.dist2 : d.dist3;
        } else if (dcode <= 15u) {
            long long basev = dcode <= 9u ? (long long)d.dist0 : (long long)d.dist1;
            long long delta = dcode <= 9u ? (long long)((dcode - 2u) >> 1) : (long long)((dcode - 8u) >> 1);
            long long r = (dcode & 1u) ? basev + delta : basev - delta;
            if (r <= 0) return ST_NON_POSITIVE_DISTANCE;
            distance = (u32)r;
        } else if (dcode <= 15u + ndirect) {
            distance = dcode - 15u;
        } else {
            u32 x = dcode - ndirect - 16u;
            u32 ndistbits = 1u + (x >> (npostfix + 1u));
            if (!in_bits(d, ndistbits, e)) return ST_EOF;
            u32 hcode = x >> npostfix;
            u32 lcode = x & ((1u << npostfix) - 1u);
            u32 offset = ((2u + (hcode & 1u)) << ndistbits) - 4u;
            distance = ((offset + e) << npostfix) + lcode + ndirect + 1u;
        }
        const u32 max_allowed = d.pos < d.window ? d.pos : d.window;
        if (dcode > 0u && distance <= max_allowed) { // :1476-1478
            d.dist3 = d.dist2; d.dist2 = d.dist1; d.dist1 = d.dist0; d.dist0 = distance;
        }
        // ---- copy_literals :1483-1542 (+ CopyLiterals state :2102-2141)
        if (distance <= max_allowed) {
            if (mlen < mb + copy_len) return ST_EXCEEDED_EXPECTED_BYTES; // :2105
            if (!out_room(d, copy_len)) return ST_OUTPUT_TOO_SMALL;
            window_copy(d, s, distance, copy_len);
            mb += copy_len;
        } else {
            if (copy_len < 4u || copy_len > 24u) return ST_INVALID_DICT_LENGTH;
            u32 wl, wb;
            if ((rc = dict_word(d, copy_len, distance - max_allowed - 1u, wl, wb))) return rc;
            if (mlen < mb + wl) return ST_EXCEEDED_EXPECTED_BYTES; // :2105 on the transformed length (Q4)
            if (!out_room(d, wl)) return ST_OUTPUT_TOO_SMALL;
            if (d.lane < wl) s.ring[(d.pos + d.lane + d.a) & RMASK] = (u8)wb;
            d.pos += wl;
            mb += wl;
            maybe_flush(d, s);
        }
        if (mb == mlen) break; // :2128
    }
    return ST_OK;
}

// Whole stream: reference decompress(), src/lib.rs:1545-2170.
FI u32 decode_stream(Dec &d, Lds &s) {
    u32 v, b;
    // parse_wbits :412-418 over the fixed tree :89-119.  Stream order: 0 -> 16; 1 nnn (n != 0) -> 17+n;
    // 1 000 mmm: m=0 -> 17, m=1 -> no entry (the reference walks off its array: UnexpectedEOF), m>=2 -> 8+m
    if (!in_bits(d, 1, b)) return ST_EOF;
    u32 wbits;
    if (!b) {
        wbits = 16;
    } else {
        if (!in_bits(d, 3, v)) return ST_EOF;
        if (v) {
            wbits = 17u + v;
        } else {
            if (!in_bits(d, 3, v)) return ST_EOF;
            if (v == 1u) return ST_EOF;
            wbits = v == 0u ? 17u : 8u + v;
        }
    }
    d.window = (1u << wbits) - 16u;
    for (;;) {
        if (++d.wd > d.wd_limit) return ST_WATCHDOG;
        u32 is_last;
        if (!in_bits(d, 1, is_last)) return ST_EOF; // parse_is_last :420
        if (is_last) {
            if (!in_bits(d, 1, b)) return ST_EOF; // parse_is_last_empty :427
            if (b) break;
        }
        if (!in_bits(d, 2, v)) return ST_EOF; // parse_m_nibbles :434
        u32 mnibbles = v == 3u ? 0u : v + 4u;
        if (mnibbles == 0u) { // metadata block (accepted with ISLAST too, Q9), :1617-1683
            if (!in_bits(d, 1, b)) return ST_EOF;
            if (b) return ST_NON_ZERO_RESERVED_BIT;
            u32 mskipbytes;
            if (!in_bits(d, 2, mskipbytes)) return ST_EOF;
            if (mskipbytes == 0u) {
                if (in_byte_tail(d)) return ST_NON_ZERO_FILL_BIT;
            } else {
                u32 skip = 0, last = 0; // parse_m_skip_len :449-467: byte << i (Q2); errors -> EOF (Q10)
                for (u32 i = 0; i < mskipbytes; i++) {
                    if (!in_bits(d, 8, last)) return ST_EOF;
                    skip |= last << i;
                }
                if (mskipbytes > 1u && last == 0u) return ST_EOF;
                skip += 1u;
                if (in_byte_tail(d)) return ST_NON_ZERO_FILL_BIT;
                if (in_remaining(d) < 8ull * skip) return ST_EOF;
                in_seek(d, d.bitpos + 8ull * skip);
            }
        } else {
            if (!in_bits(d, 4u * mnibbles, v)) return ST_EOF; // parse_m_len :469-483
            if (mnibbles > 4u && (v >> ((mnibbles - 1u) * 4u)) == 0u) return ST_NON_ZERO_TRAILER_NIBBLE;
            u32 mlen = v + 1u;
            u32 uncompressed = 0;
            if (!is_last && !in_bits(d, 1, uncompressed)) return ST_EOF; // :1689-1699
            if (uncompressed) { // :1701-1734
                if (in_byte_tail(d)) return ST_NON_ZERO_FILL_BIT;
                if (in_remaining(d) < 8ull * mlen) return ST_EOF;
                if (!out_room(d, mlen)) return ST_OUTPUT_TOO_SMALL;
                const u8 *src = (const u8 *)d.in_words + (d.bitpos >> 3);
                for (u32 done = 0; done < mlen; done += 64u) {
                    u32 n = mlen - done < 64u ? mlen - done : 64u;
                    if (d.lane < n) s.ring[(d.pos + d.lane + d.a) & RMASK] = src[done + d.lane];
                    d.pos += n;
                    maybe_flush(d, s);
                }
                in_seek(d, d.bitpos + 8ull * mlen);
            } else {
                u32 rc = compressed_meta_block(d, s, mlen);
                if (rc) return rc;
            }
        }
        if (is_last) break; // MetaBlockEnd :2146-2153
    }
    // StreamEnd :2155-2167
    if (in_byte_tail(d)) return ST_NON_ZERO_TRAILER_BIT;
    if (d.bitpos < d.bitend) return ST_EXPECTED_END_OF_STREAM;
    return ST_OK;
}

__global__ __launch_bounds__(BRX_WAVE) void brx_decode_kernel(BrxKernelArgs a) {
    __shared__ Lds s;
    Dec d;
    d.lane = threadIdx.x;
    if (a.debug_stop == 1u) return;
    d.t_dict = a.t.dict;
    d.t_xforms = a.t.xforms;
    d.scratch = a.scratch + (size_t)blockIdx.x * BRX_SCRATCH_WORDS;
    d.v_ins = d.lane < 24u ? K_INS[d.lane] : 0u;
    d.v_copy = d.lane < 24u ? K_COPY[d.lane] : 0u;
    d.v_blen = d.lane < 26u ? K_BLEN[d.lane] : 0u;
    d.v_lut0 = ((const u32 *)a.t.context_lut)[d.lane];
    d.v_lut1 = ((const u32 *)a.t.context_lut)[64u + d.lane];
    d.v_lut2 = ((const u32 *)a.t.context_lut)[128u + d.lane];
    if (a.debug_stop == 2u) return;
    for (;;) {
        // Work queue.  Every lane executes the atomic (only lane 0 adds): a lane-0-only branch here sits right
        // behind the lane-0-only status store that ends the previous iteration, and LLVM threads lanes 1..63
        // around both across the back edge -- they then spin in their own loop and never meet lane 0 again.
        u32 sid = rdl(atomicAdd(a.work_counter, d.lane == 0u ? 1u : 0u), 0);
        if (sid >= a.n) break;
        if (a.debug_stop == 3u) { if (d.lane == 0u) { a.status[sid] = 100; a.out_len[sid] = 0; } continue; }
        const u64 i0 = a.in_off[sid], i1 = a.in_off[sid + 1u];
        const u64 o0 = a.out_off[sid], o1 = a.out_off[sid + 1u];
        const u8 *inp = a.in + i0;
        const u32 mis = (u32)((uintptr_t)inp & 3u);
        d.in_words = (const u32 *)(inp - mis);
        const u64 in_len = i1 - i0;
        d.w_end = (u32)((mis + in_len + 3u) >> 2);
        d.bitend = 8ull * (mis + in_len);
        d.cbase = 0;
        d.chunkA = in_load_chunk(d, 0);
        d.chunkB = in_load_chunk(d, 64u);
        in_seek(d, 8ull * mis);
        if (a.debug_stop == 4u) { if (d.lane == 0u) { a.status[sid] = 101; a.out_len[sid] = (u64)d.win; } continue; }
        d.out = a.out + o0;
        const u64 capacity = o1 - o0;
        d.cap = capacity > 0xffffff00ull ? 0xffffff00u : (u32)capacity;
        d.pos = 0;
        d.a = (u32)((uintptr_t)d.out & 15u);
        d.vfl = d.a;
        d.dist0 = 4; d.dist1 = 11; d.dist2 = 15; d.dist3 = 16; // src/lib.rs:408
        d.needed = 0;
        d.wd = 0;
        d.wd_limit = 8ull * in_len + (u64)d.cap + 65536ull;
        d.lds_top = 0;
        d.scr_top = 0;
        u32 st = decode_stream(d, s);
        if (d.vfl < d.pos + d.a) flush_range(d, s, d.vfl, d.pos + d.a); // drain the ring
        if (d.lane == 0u) {
            a.status[sid] = (int)st;
            a.out_len[sid] = st == ST_OUTPUT_TOO_SMALL ? (u64)d.needed : (u64)d.pos;
        }
    }
}

void brx_launch_decode(const BrxKernelArgs &args, unsigned grid, void *hip_stream) {
    hipLaunchKernelGGL(brx_decode_kernel, dim3(grid), dim3(BRX_WAVE), 0, (hipStream_t)hip_stream, args);
}
