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
//   * the last 2 KiB of output live in an LDS ring (the sliding window of src/ringbuffer/); literals,
//     LZ77 copies and transformed dictionary words are produced into the ring by all 64 lanes and leave
//     for HBM in address-aligned 1 KiB blocks of 16 B per lane; back-references older than the ring are
//     read back from the stream's own HBM output          (reference: copy_literals, src/lib.rs:1483-1542)
//   * static dictionary + 121 transforms                 (reference: src/dictionary, src/transformation)
//
// The kernel is a dispatcher over out-of-line segments that hand the decoder state through LDS (Lds::st, Lds::mbw):
//   seg_frame         stream / meta-block framing, uncompressed and metadata blocks
//   cold_header       everything between MLEN and the first command: block types, context maps, prefix codes
//   generic_commands  the command loop in C++: every meta-block shape, exact end-of-input / error / capacity rules;
//                     re-entrant at three resume points (HC_*), which makes it the safety net of ...
//   asm_commands      ... the hand-written gfx950 command loop (brx_hot.S), where the time is spent
//   seg_finish        final flush
//
// Everything observable follows the reference, quirks Q1..Q15 of SURVEY.md section 2.3 included.
// There is no MFMA here: the path is integer/byte work bounded by HBM bandwidth (copy-dominated streams) and by the
// serial prefix-code chain of each stream (instruction issue rate / dependent latency), see DESIGN.md section 6.
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
    ST_OUTPUT_TOO_SMALL = 25, ST_REF_PANIC = 26, ST_WATCHDOG = 27,
    // kernel-internal, never stored: UnexpectedEOF that does NOT come from running out of bits -- the reference maps a few format
    // errors to it (Q10: a bad MSKIPLEN, src/lib.rs:460-466; the WBITS pattern without an entry; an unassigned codeword where a block
    // count is read, :977).  The resumable decode takes an item back when it meets the end of the RESIDENT input (ST_EOF) and asks for
    // more; these stay errors however much input there is (the host doubled its window up to 256 MiB for them).  Stored
    // as ST_EOF (final_status).
    ST_EOF_FORMAT = 29
};
__device__ __attribute__((always_inline)) inline unsigned final_status(unsigned st) { return st == ST_EOF_FORMAT ? (unsigned)ST_EOF : st; }

enum { LK_OK = 0, LK_NONE = 1, LK_EOF = 2 };

#ifdef BRX_SMALL
struct __attribute__((aligned(16))) Lds { // the lean instance (brx_device.h): ring, 512 words of tables, code-length scratch
    u8 ring[BRX_RING_BYTES];
    u32 tm[BRX_TM_WORDS];
    u8 lens[BRX_LENS_BYTES];         // code lengths of one alphabet (<= 704) + build_code's histogram; during the command loop:
                                     // context id -> literal tree handle (64 x u16)
    u8 trash[64];                    // per-lane dump for predicated-off LDS byte stores (see ring_put)
    u32 pad[48];
};
#elif BRX_LEVEL >= 4
struct __attribute__((aligned(16))) Lds { // (level 4: the table memory last -- brx_device.h; brx_hot.S, LDS_TM_LAST)
    u8 ring[BRX_RING_BYTES];
    u8 lens[BRX_LENS_BYTES - 512u];
    u32 st[48];
    u32 mbw[48];
    u32 pad[16];
    u8 trash[64];
    u32 tm[BRX_TM_WORDS];
};
#else
struct __attribute__((aligned(16))) Lds {
    u8 ring[BRX_RING_BYTES];
    u32 tm[BRX_TM_WORDS];
    u8 lens[BRX_LENS_BYTES - 512u]; // code lengths of one alphabet (<= 704), rounded up for 4-byte clears; during the
                                     // command loop brx_hot.S keeps its literal-context tables here
    u32 st[48];                      // decoder state parked here across calls into the cold (out-of-line) parts
    u32 mbw[48];                     // meta-block header results handed from cold_header to the command loop
    u32 pad[16];                     // bring-up counters (BRX_DEBUG_STATS)
    u8 trash[64];                    // per-lane dump for predicated-off LDS byte stores (see ring_put)
};
#endif

static_assert(sizeof(Lds) == BRX_LDS_BYTES, "Lds layout");
#if BRX_LEVEL == 0 && !defined(BRX_SMALL)
static_assert(__builtin_offsetof(Lds, st) / 4u + 3u == BRX_RESUME_CURSOR_WORD, "the host reads the parked input cursor at this word");
#endif
// One Lds per workgroup (= per wave).  File scope, so the out-of-line segments address it as LDS directly (a
// generic Lds* parameter would turn every access into a flat_* instruction).
__shared__ Lds g_lds;

#define FI __device__ __attribute__((always_inline)) inline

// Bring-up (BRX_BRINGUP=1 at build time): cycle timers of the phases of a stream, 32 slots per wave in an LDS array of their
// own (128 B more per wave: 15 waves per CU instead of 16 -- a measuring build).  PT_BEGIN(t); ... PT_ADD(slot, t) adds the
// cycles since the last mark to the slot (and counts the visit in slot + 16) and moves the mark.
#ifdef BRX_BRINGUP
__shared__ unsigned long long g_prof[32];
#define PT_BEGIN(t) unsigned long long t = __builtin_readcyclecounter()
#define PT_ADD(slot, t) do { const unsigned long long n_ = __builtin_readcyclecounter(); \
        if (threadIdx.x == 0u) { g_prof[slot] += n_ - t; g_prof[(slot) + 16] += 1ull; } t = __builtin_readcyclecounter(); } while (0)
#else
#define PT_BEGIN(t) do { } while (0)
#define PT_ADD(slot, t) do { } while (0)
#endif
// slots: 0 seg_frame  1 cold_header  2 generic_commands(HC_START / whole)  3 generic_commands(resume)  4 asm_commands  5 seg_finish
//        6 header: kind + simple code fields  7 header: 18-entry code-length code  8 header: code-length symbols  9 build_code
//        10 context map symbols  11 inverse move-to-front  12 header: everything else (block types / counts / misc / entry)
//        13 dec_load / dec_load_in  14 stream setup in the dispatcher  15 status store

// ---- wave-level primitives ---------------------------------------------------------------------------
FI u32 rfl(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
FI u32 rdl(u32 v, u32 lane) { return (u32)__builtin_amdgcn_readlane((int)v, (int)lane); }
FI u64 ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// vec with lane `lane` replaced by the wave-uniform `val` (a compare and a select: gfx9's v_writelane_b32 takes one SGPR
// operand only, its lane select would have to travel through M0)
FI u32 wrl(u32 val, u32 lane, u32 vec) { return threadIdx.x == lane ? val : vec; }

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
    u32 nav, hb_lastw, hb_lastmask; // header-path reader (hb_*): valid bits in `win`, the stream's last dword and its real bits
    // output side
    u8 *out;             // stream's output base
    __amdgpu_buffer_rsrc_t out_rsrc; // same, as a buffer resource: far back-references are buffer_load_ubyte
                                     // (a plain pointer load next to the LDS ring read gets merged into ONE flat
                                     // load by LLVM, and that trips a backend bug in non-kernel functions)
    u8 *mirror;          // nullptr, or the same slot in host-visible (pinned, mapped) memory: every store of output
    __amdgpu_buffer_rsrc_t out2_rsrc; // bytes goes there too (the D2H copy fused into the decode); null resource = dropped
    u32 cap;             // capacity (clamped to < 2^32-64)
    u32 pos;             // bytes produced (reference: Decompressor.count_output)
    u32 a;               // (uintptr_t)out & 15: ring/global 16-B alignment skew
    u32 vfl;             // flushed, in skewed coordinates (pos + a)
    u32 window;          // (1<<WBITS)-16
    u32 dist0, dist1, dist2, dist3; // last distances, dist0 most recent
    // table memory
    u32 lds_top, scr_top;
    u32 need_cur, need_peak;   // words of table memory allocated right now / at this meta-block's peak: the level that holds it
    u32 *scratch;              // this stream's spill slab, nullptr until the first spill (scratch_claim)
    const BrxSlabPool *pool;
    // per-lane constant vectors
    u32 v_ic; // lanes 0..23: insert length codes, lanes 32..55: copy length codes ((base << 5) | extra bits)
    u32 v_lut0, v_lut1, v_lut2;
    u32 needed; // for ST_OUTPUT_TOO_SMALL
    u64 wd, wd_limit; // loop watchdog: every command / meta-block consumes a bit or emits a byte
    const u8 *t_dict;
    const BrxTransform *t_xforms;
    const u32 *t_lut; // Lut0 | Lut1 | Lut2 as dwords
};


// ---- table memory: LDS first, HBM spill beyond ---------------------------------------------------------
// An object lives entirely in LDS (word address < BRX_TM_WORDS) or entirely in the HBM spill arena.  The
// accessors take the address space as a template flag chosen by ONE wave-uniform branch per object, so
// the compiler emits real ds_* / global_* instructions (a per-access select makes it fall back to
// flat_load, which took the v1 kernel to ~4 flat loads per symbol -- profiles/r01a_pmc.csv).
template <bool INL> FI u32 tm_ld32(const Dec &d, const Lds &s, u32 wa) {
    if (INL) return s.tm[wa];
    return __builtin_nontemporal_load(&d.scratch[wa - BRX_TM_WORDS]); // nt: keeps LLVM from merging the two
                                                                    // address spaces into one flat access
}
template <bool INL> FI void tm_st32(const Dec &d, Lds &s, u32 wa, u32 v) {
    if (INL) s.tm[wa] = v; else __builtin_nontemporal_store(v, &d.scratch[wa - BRX_TM_WORDS]);
}
template <bool INL> FI u32 tm_ld16(const Dec &d, const Lds &s, u32 ha) { // halfword address
    if (INL) return ((const u16 *)s.tm)[ha];
    return __builtin_nontemporal_load(&((const u16 *)d.scratch)[ha - BRX_TM_WORDS * 2]);
}
template <bool INL> FI void tm_st16(const Dec &d, Lds &s, u32 ha, u32 v) {
    if (INL) ((u16 *)s.tm)[ha] = (u16)v; else __builtin_nontemporal_store((u16)v, &((u16 *)d.scratch)[ha - BRX_TM_WORDS * 2]);
}
template <bool INL> FI u32 tm_ld8(const Dec &d, const Lds &s, u32 ba) {
    if (INL) return ((const u8 *)s.tm)[ba];
    return __builtin_nontemporal_load(&((const u8 *)d.scratch)[ba - TM_BYTES]);
}
template <bool INL> FI void tm_st8(const Dec &d, Lds &s, u32 ba, u32 v) {
    if (INL) ((u8 *)s.tm)[ba] = (u8)v; else __builtin_nontemporal_store((u8)v, &((u8 *)d.scratch)[ba - TM_BYTES]);
}
// wave-uniform scalar reads (address uniform): dispatch on the address itself
FI u32 tm_u32(const Dec &d, const Lds &s, u32 wa) {
    return wa < BRX_TM_WORDS ? rfl(tm_ld32<true>(d, s, wa)) : rfl(tm_ld32<false>(d, s, wa));
}
FI u32 tm_u8(const Dec &d, const Lds &s, u32 ba) {
    return ba < TM_BYTES ? rfl(tm_ld8<true>(d, s, ba)) : rfl(tm_ld8<false>(d, s, ba));
}
FI void tm_set8(const Dec &d, Lds &s, u32 ba, u32 v) { // uniform address, lane 0 stores
    if (ba < TM_BYTES) { if (d.lane == 0u) tm_st8<true>(d, s, ba, v); }
    else { if (d.lane == 0u) tm_st8<false>(d, s, ba, v); }
}
FI void tm_set32(const Dec &d, Lds &s, u32 wa, u32 v) {
    if (wa < BRX_TM_WORDS) { if (d.lane == 0u) tm_st32<true>(d, s, wa, v); }
    else { if (d.lane == 0u) tm_st32<false>(d, s, wa, v); }
}
FI void tm_zero_words(const Dec &d, Lds &s, u32 wa, u32 n) {
    if (wa < BRX_TM_WORDS) { for (u32 k = d.lane; k < n; k += 64u) tm_st32<true>(d, s, wa + k, 0u); }
    else { for (u32 k = d.lane; k < n; k += 64u) tm_st32<false>(d, s, wa + k, 0u); }
}
FI void tm_zero_bytes(const Dec &d, Lds &s, u32 ba, u32 n) {
    if (ba < TM_BYTES) { for (u32 k = d.lane; k < n; k += 64u) tm_st8<true>(d, s, ba + k, 0u); }
    else { for (u32 k = d.lane; k < n; k += 64u) tm_st8<false>(d, s, ba + k, 0u); }
}
// Spill slabs: claimed from the context's pool the first time a stream needs one, released when the stream ends
// (scratch_release in the dispatcher).  Wave-uniform; lane 0 does the atomics.  A wave that finds every slab taken
// waits for one: holders never wait for anything, so this cannot deadlock.
FI u32 *scratch_claim(const BrxSlabPool *pool) {
    const u32 nwords = pool->count >> 5;
    u32 w = (blockIdx.x * 7u) % nwords;
    const u32 nwords0 = w == 0u ? nwords : w; // the word that ends the first pass: the one in front of the start
    bool waited = false;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        // (the pool has a slab for every wave of every launch in flight on the context -- brx_api.cpp, pool_need; until round 6 it
        // followed the largest single grid, and a wave of a second, overlapping launch could wait for as long as the other one's
        // slab-class streams took -- so nobody waits here.  A pool that stays exhausted for 4 s -- 100 MHz counter -- is a bug: the
        // stream gets the watchdog status instead of the device a hang.)
        if (__builtin_amdgcn_s_memrealtime() - t0 > 400000000ull) return nullptr;
        u32 got = 0xffffffffu;
        if (threadIdx.x == 0u) {
            u32 cur = __hip_atomic_load(&pool->bitmap[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (cur != 0xffffffffu) { // (a bit lost to another wave is set in what the atomic returns: at most 32 turns)
                const u32 bit = (u32)__builtin_ctz(~cur);
                const u32 old = atomicOr(&pool->bitmap[w], 1u << bit);
                if (((old >> bit) & 1u) == 0u) { got = w * 32u + bit; break; }
                cur = old;
            }
        }
        got = rfl(got);
        if (got != 0xffffffffu) return pool->slabs + (size_t)got * BRX_SCRATCH_WORDS;
        if (!waited && w + 1u == nwords0) { // (one pass over the bitmap found nothing)
            waited = true;
            if (threadIdx.x == 0u) (void)atomicAdd(pool->waits, 1u);
        }
        w = w + 1u == nwords ? 0u : w + 1u;
        __builtin_amdgcn_s_sleep(8);
    }
}
FI void scratch_release(const BrxSlabPool *pool, const u32 *slab) {
    if (slab == pool->sink) return; // (never claimed)
    const u32 idx = (u32)((size_t)(slab - pool->slabs) / BRX_SCRATCH_WORDS);
    if (threadIdx.x == 0u) atomicAnd(&pool->bitmap[idx >> 5], ~(1u << (idx & 31u)));
}
// Objects never straddle the LDS / HBM boundary.
#ifdef BRX_SMALL
// The lean instance has no slab: an object that does not fit sets the overflow mark (Dec::scr_top != 0) and gets address 0 --
// whatever is then written there stays inside the wave's own LDS (the largest object, a 704-symbol code of 32-bit entries,
// is 736 words; behind tm[] lie 3 072 bytes of the same wave), and the caller lists the stream for the regular kernel.
FI u32 tm_alloc(Dec &d, u32 nwords) {
    if (d.lds_top + nwords <= BRX_TM_WORDS) {
        const u32 r = d.lds_top;
        d.lds_top += nwords;
        return r;
    }
    d.scr_top = 1u;
    return 0u;
}
#else
FI u32 tm_alloc(Dec &d, u32 nwords) {
    // (what a wider level needs for the same objects: it packs them one behind the other, without the hole this level leaves at
    // the end of its LDS part when an object does not fit)
    d.need_cur += nwords;
    d.need_peak = d.need_cur > d.need_peak ? d.need_cur : d.need_peak; // (0xffffffff = "no slab could be had" stays)
    if (d.lds_top + nwords <= BRX_TM_WORDS) {
        u32 r = d.lds_top;
        d.lds_top += nwords;
        return r;
    }
    if (d.scratch == nullptr) d.scratch = scratch_claim(d.pool);
    if (d.scratch == nullptr) { // no slab: the tables go to the pool's sink (shared, so they are garbage), the header ends with ST_WATCHDOG
        d.need_peak = 0xffffffffu;
        d.scratch = d.pool->sink;
    }
    u32 r = BRX_TM_WORDS + d.scr_top;
    d.scr_top += nwords;
    return r;
}
#endif

// ---- bit input (reference: src/bitreader/mod.rs:21-303) -------------------------------------------------
// Branch-free on purpose (clamped index + select): a per-lane branch here makes LLVM's uniformity analysis
// treat the scalar reader state that is merged at the same join block (cbase, the bit window ...) as
// divergent, which moved the WHOLE command loop state into VGPRs (callers guarantee w_end >= 1).
FI u32 in_load_chunk(const Dec &d, u32 c) {
    u32 i = c + d.lane;
    u32 j = i < d.w_end ? i : d.w_end - 1u;
    u32 v = d.in_words[j];
    return i < d.w_end ? v : 0u;
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

// Lds::st slots beyond the Dec fields (dec_store uses 0..32)
#define ST_STARTED 33 // 0 before the stream header (WBITS) has been read
#define ST_ISLAST 34  // ISLAST of the meta-block handed to the command loop
#define ST_MLEN 35    // its MLEN
#define ST_IACTAB 36  // (2 words) BrxDeviceTables::iac for the assembly loop
#define ST_POOL 38    // (2 words) const BrxSlabPool *
#define ST_MIRROR 40  // (2 words) host-visible mirror of the output slot, or 0
#define ST_PAUSE_AT 43 // (2 words) resumable mode: seg_frame stops between two meta-blocks once this many bytes are out (else ~0)
#define ST_IN_LOW 45   // (2 words) ... or once the input cursor is this far (BrxResume::in_low; else ~0)
#define ST_SPEC 47    // bit 1: option command_loop = 6 (no meta-block qualifies for the assembly loop); bit 0 = 1: batch decode -- the assembly loop may run past the end of the input (a meta-block that did is taken back and decoded
                      // again by the C++ loop: "speculative end" in the kernel's stream loop); 0: the resumable decode, END_MARGIN applies
#define ST_NEED 42    // after a header whose tables spilled: words of table memory the meta-block needs (Dec::need_peak)
#define SEG_NEED_HEADER 100u // seg_frame: a compressed meta-block follows (anything < 100 is a final status)
#define SEG_PAUSED 101u      // seg_frame (resumable mode): stopped between two meta-blocks, ST_PAUSE_AT reached
// generic_commands modes / return value, and the Lds::mbw slots that carry a parked command
#define HC_WHOLE 0u      // run the whole meta-block
#define HC_START 1u      // first insert&copy symbol only, then park at R1
#define HC_RESUME_R0 2u  // resume: insert&copy symbol due; one command, park
#define HC_RESUME_R1 3u  // resume at the loop top; one command, park
#define HC_RESUME_R2 4u  // resume with the distance known; one command, park
#define HC_RESUME_R1_WHOLE 5u // resume at the loop top and run the meta-block to its end
#define HC_TO_END 8u     // flag on HC_RESUME_R0 .. R2: do not park after one command, run the meta-block to its end (the assembly
                         // loop reported the cursor within the last dwords of the stream: it would only hand straight back)
#define HC_CONTINUE 200u // returned when parked (meta-block not finished)
#define MBW_ASM 31
#define MBW_MBLEFT 32
#define MBW_INS 33
#define MBW_CPY 34
#define MBW_IZ 35
#define MBW_DIST 36
#define MBW_DISTBAD 37
#define MBW_EXIT 38
#define BRX_END_MARGIN 5u          // = END_MARGIN of brx_hot.S: what the loop may consume between a poisoned refill and its next exit test
#ifndef BRX_SPEC_CK_DWORDS
#define BRX_SPEC_CK_DWORDS 64u     // speculative end: a long stream's loop is first poisoned this far in front of the end of the input (the checkpoint)
#endif
#ifndef BRX_SPEC_CK_MIN_DWORDS
#define BRX_SPEC_CK_MIN_DWORDS 1024u // ... if it is entered with more input than this left; shorter ones take their checkpoint at the first entry
#endif
#define MBW_WSAFE 39 // the dword at which the assembly loop is poisoned (brx_hot.S, .Lspecial): set by the dispatcher before every call

#ifndef BRX_SMALL
// ---- parking the decoder state in LDS ------------------------------------------------------------------
// The cold parts of the decoder (meta-block header parsing with its code builders; the table-memory command
// loop for oversized meta-blocks) are real out-of-line functions so that their register needs do not leak
// into the register allocation of the hot command loop.  State crosses the call through Lds::st.
FI void put64(Lds &s, u32 i, u64 v) { s.st[i] = (u32)v; s.st[i + 1u] = (u32)(v >> 32); }
FI u64 get64(const Lds &s, u32 i) { return (u64)rfl(s.st[i]) | ((u64)rfl(s.st[i + 1u]) << 32); }
FI void dec_store(const Dec &d, Lds &s) {
    { // every lane stores the same wave-uniform values: no lane-0 branch
        put64(s, 0, (u64)(uintptr_t)d.in_words); s.st[2] = d.w_end; put64(s, 3, d.bitpos); put64(s, 5, d.bitend);
        put64(s, 7, (u64)(uintptr_t)d.out); s.st[9] = d.cap; s.st[10] = d.pos; s.st[11] = d.a; s.st[12] = d.vfl;
        s.st[13] = d.window; s.st[14] = d.dist0; s.st[15] = d.dist1; s.st[16] = d.dist2; s.st[17] = d.dist3;
        s.st[18] = d.lds_top; s.st[19] = d.scr_top; put64(s, 20, (u64)(uintptr_t)d.scratch); s.st[22] = d.needed;
        put64(s, 23, (u64)(uintptr_t)d.t_dict); put64(s, 25, (u64)(uintptr_t)d.t_xforms);
        put64(s, 27, (u64)(uintptr_t)d.t_lut); put64(s, 29, d.wd); put64(s, 31, d.wd_limit);
        put64(s, ST_POOL, (u64)(uintptr_t)d.pool);
        put64(s, ST_MIRROR, (u64)(uintptr_t)d.mirror);
    }
}
FI void dec_load(Dec &d, const Lds &s) {
    d.lane = threadIdx.x;
    d.in_words = (const u32 *)(uintptr_t)get64(s, 0); d.w_end = rfl(s.st[2]);
    d.bitend = get64(s, 5);
    d.out = (u8 *)(uintptr_t)get64(s, 7); d.cap = rfl(s.st[9]);
    d.out_rsrc = __builtin_amdgcn_make_buffer_rsrc(d.out, 0, d.cap, 0x00020000); // range-checked against the capacity
    d.mirror = (u8 *)(uintptr_t)get64(s, ST_MIRROR);
    d.out2_rsrc = __builtin_amdgcn_make_buffer_rsrc(d.mirror, 0, d.mirror ? d.cap : 0u, 0x00020000);
    d.pos = rfl(s.st[10]); d.a = rfl(s.st[11]);
    d.vfl = rfl(s.st[12]); d.window = rfl(s.st[13]);
    d.dist0 = rfl(s.st[14]); d.dist1 = rfl(s.st[15]); d.dist2 = rfl(s.st[16]); d.dist3 = rfl(s.st[17]);
    d.lds_top = rfl(s.st[18]); d.scr_top = rfl(s.st[19]); d.scratch = (u32 *)(uintptr_t)get64(s, 20);
    d.needed = rfl(s.st[22]);
    d.t_dict = (const u8 *)(uintptr_t)get64(s, 23); d.t_xforms = (const BrxTransform *)(uintptr_t)get64(s, 25);
    d.t_lut = (const u32 *)(uintptr_t)get64(s, 27); d.wd = get64(s, 29); d.wd_limit = get64(s, 31);
    d.pool = (const BrxSlabPool *)(uintptr_t)get64(s, ST_POOL);
    d.need_cur = 0u; d.need_peak = 0u;
    // (the per-lane constant vectors v_ic / v_lut0..2 are not loaded here: the kernel loads them once per wave and hands them
    // to generic_commands as arguments -- five global loads less per call of a segment)
    d.cbase = 0xffffff00u; // force a re-stage of the input chunks
    d.chunkA = 0; d.chunkB = 0;
    in_seek(d, get64(s, 3));
}

// The header path (cold_header) needs the input cursor and the table memory, nothing of the output side: loading and
// storing only those keeps ~30 wave-uniform values out of its register allocation (with all of Dec live from dec_load to
// dec_store LLVM spilled SGPRs into VGPR lanes all over the header code: 187 spill writes / 412 reloads, r02 build).
FI void dec_load_in(Dec &d, const Lds &s) {
    d.lane = threadIdx.x;
    d.in_words = (const u32 *)(uintptr_t)get64(s, 0); d.w_end = rfl(s.st[2]);
    d.bitend = get64(s, 5);
    d.lds_top = rfl(s.st[18]); d.scr_top = rfl(s.st[19]); d.scratch = (u32 *)(uintptr_t)get64(s, 20);
    d.pool = (const BrxSlabPool *)(uintptr_t)get64(s, ST_POOL);
    d.need_cur = 0u; d.need_peak = 0u;
    d.cbase = 0xffffff00u; // force a re-stage of the input chunks
    d.chunkA = 0; d.chunkB = 0;
    in_seek(d, get64(s, 3));
}
FI void dec_store_in(const Dec &d, Lds &s) {
    put64(s, 3, d.bitpos);
    s.st[18] = d.lds_top; s.st[19] = d.scr_top; put64(s, 20, (u64)(uintptr_t)d.scratch);
    s.st[ST_NEED] = d.need_peak;
}

#endif // !BRX_SMALL

// ---- prefix codes ----------------------------------------------------------------------------------------
// Table layout in table memory (word address h): BRX_HDR_WORDS = 17 header words, then the symbols in (length, symbol)
// order -- as u16 (literal, insert&copy, block-type / block-count / context-map codes) or as u32 (distance codes, WIDE).
//   header[L], L = 1..15 : limit[L] << 16 | (base[L] & 0xffff)
//                          limit[L] = (first_code[L] + count[L]) << (15-L) -- the exclusive upper bound of the length-L codes as
//                          a left-aligned 15-bit value (2^15 = the complete code's last bound);
//                          base[L] = offset[L] - first_code[L] (two's complement, it fits 16 bits): code value + base = symbol index
//   header[0]            : 0 (never matches)
//   header[16]           : kind | max_len << 8 | x << 16  (kind 0 empty, 1 one symbol x, 2 general with x symbols)
// (Round 4: limit and base share a word -- 17 header words instead of 32 per tree, which is what keeps the tables of real text
// at quality 10 / 11 inside the regular kernel's table memory: lcet10.txt 2 208 -> 1 7xx words.  The compare takes the word as it
// is against (window << 16 | 0xffff); the assembly loop splits a tree's words once when it loads them.)
// One v_cmp of the bit-reversed window against the limits gives the code length (lowest matching lane).
// Lookup = reference Tree::lookup_symbol (src/huffman/tree/mod.rs:63-93): zero bits for a single-symbol
// code (Q5); an unassigned codeword of an incomplete code reads max_len+1 bits and yields None (Q15).
//
// Symbol entries hold the plain symbol value when the tables are built.  For a meta-block that qualifies for the
// assembly loop, prepare_fast_tables() rewrites them in place into the form that loop wants (MBW_ASM = 1; the C++ loop
// reads the same form then):
//   literal trees      sym | info << 8, info = the literal's share of the NEXT literal's context id (context_info())
//   insert&copy trees  sym << 4 = byte offset of the symbol's record in BrxDeviceTables::iac
//   distance trees     0x80000000 | code for the 16 last-distance codes, else nbits | base << 5 with
//                      distance = base + (extra << NPOSTFIX)  (decode_distance, src/lib.rs:1412-1481)
#define BRX_HDR_WORDS 17u
#define BRX_HDR_INFO 16u // word of the header that holds kind | max_len << 8 | x << 16
#define BRX_DIST_UNFIT 0xc0000000u
template <bool INL, bool WIDE> FI u32 decode_sym_as(Dec &d, const Lds &s, u32 h, u32 &sym) {
    u32 hv = tm_ld32<INL>(d, s, h + (d.lane & 31u)); // lanes 1..15: limit << 16 | base, lane 16: the info word
    u32 h0 = rdl(hv, BRX_HDR_INFO);
    u32 kind = h0 & 3u;
    if (kind == 0u) return LK_NONE;
    if (kind == 1u) {
        sym = h0 >> 16;
        if (WIDE && sym == 0xffffu) sym = rfl(tm_ld32<INL>(d, s, h + 13u)); // (prepare_fast_tables: a one-symbol explicit-distance code)
        return LK_OK;
    }
    u64 rem = in_remaining(d);
    u32 peek = in_peek_raw(d) & 0x7fffu;
    if (rem < 15u) peek &= (1u << (u32)rem) - 1u;
    u32 v = __brev(peek) >> 17; // first stream bit = MSB of a 15-bit left-aligned code
    u64 m = ballot(((v << 16) | 0xffffu) < hv) & 0xfffeull; // lanes 1 .. 15
    if (m == 0ull) {
        u32 maxlen = (h0 >> 8) & 0xffu;
        return rem >= (u64)(maxlen + 1u) ? LK_NONE : LK_EOF;
    }
    u32 L = (u32)__builtin_ctzll(m);
    if ((u64)L > rem) return LK_EOF;
    u32 base = rdl(hv, L); // (its low half; the sum is taken modulo 2^16)
    u32 idx = ((v >> (15u - L)) + base) & 0xffffu;
    sym = WIDE ? rfl(tm_ld32<INL>(d, s, h + BRX_HDR_WORDS + idx)) : rfl(tm_ld16<INL>(d, s, (h + BRX_HDR_WORDS) * 2u + idx));
    in_consume(d, L);
    return LK_OK;
}
FI u32 decode_sym(Dec &d, const Lds &s, u32 h, u32 &sym) {
    if (h < BRX_TM_WORDS) return decode_sym_as<true, false>(d, s, h, sym);
    return decode_sym_as<false, false>(d, s, h, sym);
}
FI u32 decode_sym_wide(Dec &d, const Lds &s, u32 h, u32 &sym) {
    if (h < BRX_TM_WORDS) return decode_sym_as<true, true>(d, s, h, sym);
    return decode_sym_as<false, true>(d, s, h, sym);
}

// ---- the header path's bit reader ----------------------------------------------------------------------------
// Everything between MLEN and the first command (cold_header) reads its bits through hb_*: a 64-bit window with >= 32
// valid bits, refilled a dword at a time, over the input PADDED WITH ZEROS behind its last real bit -- and never checks
// for the end of the input per field.  One check at the end (and at every error exit) restores the reference's
// behaviour exactly: each read of the header fails in exactly one way when the bits run out -- UnexpectedEOF (every
// `?` on a bit read in src/lib.rs:501-1177, the Ok(None) arms of the lookups) -- and a prefix-code lookup is decided by the
// bits it consumes, never by the ones it only peeks at.  So up to the first read that crosses the end both readers see the
// same bits and make the same decisions; the exact reader then stops with UnexpectedEOF; this one runs on over zeros
// (every loop of the header is bounded by an alphabet size or a declared count) and finds cursor > end afterwards.
#ifdef BRX_SMALL
// (the lean instance: the whole input -- at most 128 dwords -- was staged once, the bits behind the stream's end masked off)
FI u32 hb_word(Dec &d, u32 w) {
    const u32 v = rfl(w < 64u ? rdl(d.chunkA, w & 63u) : rdl(d.chunkB, w & 63u));
    return w < 128u ? v : 0u;
}
#else
FI u32 hb_word(Dec &d, u32 w) {
    const u32 v = in_word(d, w); // (0 from the stream's last dword on)
    return w == d.hb_lastw ? v & d.hb_lastmask : v;
}
#endif
FI void hb_begin(Dec &d) { // from the exact reader's cursor d.bitpos
    const u32 w = (u32)(d.bitpos >> 5), sh = (u32)d.bitpos & 31u;
    d.hb_lastw = (u32)((d.bitend - 1ull) >> 5);
    const u32 r = (u32)d.bitend & 31u;
    d.hb_lastmask = r ? (1u << r) - 1u : 0xffffffffu;
    const u64 lo = hb_word(d, w), hi = hb_word(d, w + 1u);
    d.win = (lo | (hi << 32)) >> sh;
    d.nav = 64u - sh;
    d.ww = w + 2u; // the next dword to enter the window
}
FI u64 hb_pos(const Dec &d) { return 32ull * d.ww - d.nav; }
FI bool hb_over(const Dec &d) { return hb_pos(d) > d.bitend; }
FI u32 hb_peek(const Dec &d) { return (u32)d.win; } // 32 valid bits
FI void hb_skip(Dec &d, u32 n) { // n <= 32
    d.win >>= n;
    d.nav -= n;
    if (d.nav < 32u) {
        d.win |= (u64)hb_word(d, d.ww) << d.nav;
        d.nav += 32u;
        d.ww += 1u;
    }
}
FI u32 hb_bits(Dec &d, u32 n) { // n <= 24
    const u32 v = hb_peek(d) & ((1u << n) - 1u);
    hb_skip(d, n);
    return v;
}

// inclusive prefix sum inside each row of 16 lanes (DPP row_shr 1 / 2 / 4 / 8, zeros shifted in)
FI u32 row_scan_add(u32 x) {
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);
    return x;
}
// lanes below this one whose bit is set in the wave mask m
FI u32 lanes_below(u64 m) { return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); }

// The header words of a general code from its length histogram: lane L (1..15) passes the number of length-L codes in
// `cnt` (other lanes: anything).  Canonical assignment, reference src/huffman/mod.rs:19-43 (the bl_count[0] quirk Q7
// vanishes under the reference's own masking of the code to `len` bits), as two prefix sums over the lanes:
//   limit[L] = (first_code[L] + count[L]) << (15 - L) = sum over k <= L of count[k] << (15 - k)     (the Kraft sum)
//   base[L]  = offset[L] - first_code[L],  offset[L] = sum over k < L of count[k]
// Allocates the table (header + `nnz` symbol entries), stores the header, returns h; off = offset[L] in lane L.
// Precondition (checked by the callers like the reference does): Kraft sum <= 1, at least 2 non-zero lengths.
FI u32 emit_code_header(Dec &d, Lds &s, u32 cnt, const bool wide, u32 &off) {
    const u32 lane = d.lane, L = lane & 15u;
    const u32 c = (lane - 1u) < 15u ? cnt : 0u;
    const u32 lim = row_scan_add(c << (15u - L));
    const u32 offi = row_scan_add(c);
    off = offi - c;
    const u32 base = off - ((lim >> (15u - L)) - c);
    const u32 nnz = rdl(offi, 15);
    const u32 maxlen = 63u - (u32)__builtin_clzll(ballot(c != 0u) | 1ull);
    const u32 h = tm_alloc(d, BRX_HDR_WORDS + (wide ? nnz : ((nnz + 1u) >> 1)));
    const u32 w = lane == 0u ? 0u : lane == BRX_HDR_INFO ? (2u | (maxlen << 8) | (nnz << 16)) : (lim << 16) | (base & 0xffffu);
    if (h < BRX_TM_WORDS) { if (lane < BRX_HDR_WORDS) tm_st32<true>(d, s, h + lane, w); }
    else { if (lane < BRX_HDR_WORDS) tm_st32<false>(d, s, h + lane, w); }
    return h;
}
FI void code_put_symbol(const Dec &d, Lds &s, u32 h, const bool wide, bool on, u32 slot, u32 sym) {
    if (wide) {
        if (h < BRX_TM_WORDS) { if (on) tm_st32<true>(d, s, h + BRX_HDR_WORDS + slot, sym); }
        else { if (on) tm_st32<false>(d, s, h + BRX_HDR_WORDS + slot, sym); }
    } else {
        if (h < BRX_TM_WORDS) { if (on) tm_st16<true>(d, s, (h + BRX_HDR_WORDS) * 2u + slot, sym); }
        else { if (on) tm_st16<false>(d, s, (h + BRX_HDR_WORDS) * 2u + slot, sym); }
    }
}

// Build a general code from s.lens[0..n).  Lane-parallel: the histogram is one LDS atomic per 64 symbols, the header two
// prefix sums (emit_code_header), and a symbol's place in the (length, symbol)-sorted list is
//   running offset of its length  +  number of lower lanes of its chunk with the same length
// (four ballots of the length's bits give every lane the mask of its peers; the running offsets live in LDS and move by one
// atomic add per chunk).  16 words behind the longest alphabet (704) serve as the histogram / the running offsets.
FI u32 build_code(Dec &d, Lds &s, u32 n, const bool wide = false) {
    const u32 lane = d.lane;
    u32 *const cnt = (u32 *)&s.lens[704];
    if (lane < 16u) cnt[lane] = 0u;
    const u32 nch = (n + 63u) >> 6;
    for (u32 c = 0; c < nch; c++) {
        const u32 i = c * 64u + lane;
        const u32 my = i < n ? s.lens[i] : 0u;
        __hip_atomic_fetch_add(&cnt[my], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // (zeros count into cnt[0]: never read)
    }
    u32 off;
    const u32 h = emit_code_header(d, s, cnt[lane & 15u], wide, off);
    if (lane < 16u) cnt[lane] = off;
    for (u32 c = 0; c < nch; c++) {
        const u32 i = c * 64u + lane;
        const u32 my = i < n ? s.lens[i] : 0u;
        if (ballot(my != 0u) == 0ull) continue;
        const u64 b0 = ballot((my & 1u) != 0u), b1 = ballot((my & 2u) != 0u), b2 = ballot((my & 4u) != 0u), b3 = ballot((my & 8u) != 0u);
        u64 eq = (my & 1u) ? b0 : ~b0;
        eq &= (my & 2u) ? b1 : ~b1;
        eq &= (my & 4u) ? b2 : ~b2;
        eq &= (my & 8u) ? b3 : ~b3;
        const u32 slot = cnt[my] + lanes_below(eq);
        code_put_symbol(d, s, h, wide, my != 0u, slot, i);
        __hip_atomic_fetch_add(&cnt[my], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return h;
}

// A simple prefix code of 2..4 symbols (src/lib.rs:597-665): the same table from the (length, symbol) pairs directly --
// no pass over the alphabet.  Within one length the symbols go in ascending order (what the reference's insertion order
// amounts to in every NSYM case), so the list is the pairs sorted by (length, symbol).
FI u32 build_simple(Dec &d, Lds &s, u32 nsym, u32 k0, u32 k1, u32 k2, u32 k3, const bool wide) { // k = length << 16 | symbol
    if (nsym < 3u) k2 = 0xffffffffu;
    if (nsym < 4u) k3 = 0xffffffffu;
    u32 t;
#define BRX_CSWAP(a, b) do { t = a < b ? a : b; b = a < b ? b : a; a = t; } while (0)
    BRX_CSWAP(k0, k1); BRX_CSWAP(k2, k3); BRX_CSWAP(k0, k2); BRX_CSWAP(k1, k3); BRX_CSWAP(k1, k2);
#undef BRX_CSWAP
    const u32 lane = d.lane;
    const u32 cnt = (u32)((k0 >> 16) == lane) + (u32)((k1 >> 16) == lane) + (u32)((k2 >> 16) == lane) + (u32)((k3 >> 16) == lane);
    u32 off;
    const u32 h = emit_code_header(d, s, cnt, wide, off);
    const u32 mine = lane == 0u ? k0 : lane == 1u ? k1 : lane == 2u ? k2 : k3;
    code_put_symbol(d, s, h, wide, lane < nsym, lane, mine & 0xffffu);
    return h;
}

FI u32 build_single(Dec &d, Lds &s, u32 sym) {
    u32 h = tm_alloc(d, BRX_HDR_WORDS);
    u32 w = d.lane == BRX_HDR_INFO ? (1u | (sym << 16)) : 0u;
    if (h < BRX_TM_WORDS) { if (d.lane < BRX_HDR_WORDS) tm_st32<true>(d, s, h + d.lane, w); }
    else { if (d.lane < BRX_HDR_WORDS) tm_st32<false>(d, s, h + d.lane, w); }
    return h;
}

FI void lens_clear(const Dec &d, Lds &s, u32 n) {
    for (u32 i = d.lane * 4u; i < n; i += 256u) *(u32 *)&s.lens[i] = 0u;
}

// parse_complex_prefix_code, src/lib.rs:667-875 (Q6, Q15): code-length code over {0..17}, then the lengths.
// Fills s.lens[0..alphabet); the caller builds the table.  (hb_* reader: no end-of-input tests here, see above.)
FI u32 read_complex_lens(Dec &d, Lds &s, u32 hskip, u32 alphabet) {
    PT_BEGIN(pt);
    const u32 lane = d.lane;
    u32 clv = 0; // lane sy < 18: length of code-length symbol sy
    u32 sum = 0, nonzero = 0, single = 0;
    // Round 6: the up to 18 symbols of the fixed code, lane-parallel.  The serial loop below costs ~6 800 cycles per prefix code (a
    // fifth of a header: profiles/r06_phases.txt); here every lane L decodes "the symbol that would start at bit L" (2 .. 4 bits),
    // the chain 0 -> 0 + len(0) -> ... is followed by pointer doubling (five shuffles) and lane k picks hop^k(0) by the bits of k
    // (five more): lane k then holds symbol k's position, length and value, a prefix sum finds where the code space fills up
    // (the reference's `sum == 32: break` / `sum > 32: error`, src/lib.rs:700-745, decided by the FIRST symbol that gets there) and
    // exactly the bits the serial loop would have consumed are skipped.  Chains that leave the 64 lanes (more than ~15 four-bit
    // symbols: lengths 1 and 5 throughout) take the serial loop.
    bool parallel_done = false;
#ifndef BRX_NO_PAR_CLCODE
    {
        const u32 w0 = hb_word(d, d.ww), w1 = hb_word(d, d.ww + 1u);
        const u64 lo = d.nav < 64u ? (d.win | ((u64)w0 << d.nav)) : d.win;
        const u64 hi = ((u64)w0 >> (64u - d.nav)) | ((u64)w1 << (d.nav - 32u));
        const u32 p = (u32)((lo >> lane) | (lane ? hi << ((64u - lane) & 63u) : 0ull)) & 15u;
        // fixed code (src/lib.rs:120-125), stream order: 00->0 01->3 10->4 110->2 1110->1 1111->5 -- selects, no per-lane branch
        const u32 len = (p & 1u) == 0u ? 2u : (p & 2u) == 0u ? 2u : (p & 4u) == 0u ? 3u : 4u;
        const u32 val = (p & 1u) == 0u ? ((p & 2u) ? 3u : 0u) : (p & 2u) == 0u ? 4u : (p & 4u) == 0u ? 2u : ((p & 8u) ? 5u : 1u);
        const u32 lv = len | (val << 4);
        const u32 nx = lane + len;
        const u32 j1 = nx < 63u ? nx : 63u;
        const u32 j2 = (u32)__shfl((int)j1, (int)j1), j4 = (u32)__shfl((int)j2, (int)j2), j8 = (u32)__shfl((int)j4, (int)j4),
                  j16 = (u32)__shfl((int)j8, (int)j8);
        u32 pos = 0; // lane k: hop^k(0) -- every lane shuffles at every step, the select keeps what its k wants
        {
            const u32 t1 = (u32)__shfl((int)j1, (int)pos); pos = (lane & 1u) ? t1 : pos;
            const u32 t2 = (u32)__shfl((int)j2, (int)pos); pos = (lane & 2u) ? t2 : pos;
            const u32 t4 = (u32)__shfl((int)j4, (int)pos); pos = (lane & 4u) ? t4 : pos;
            const u32 t8 = (u32)__shfl((int)j8, (int)pos); pos = (lane & 8u) ? t8 : pos;
            const u32 t16 = (u32)__shfl((int)j16, (int)pos); pos = (lane & 16u) ? t16 : pos;
        }
        const u32 mine = (u32)__shfl((int)lv, (int)pos); // symbol k = lane: len | val << 4, at bit `pos`
        const u32 n = 18u - hskip;
        const u32 myval = lane < n ? mine >> 4 : 0u, mylen = mine & 15u;
        const u32 space = myval ? 32u >> myval : 0u;
        const u32 sc = row_scan_add(space);
        const u32 cum = lane >= 16u ? sc + rdl(sc, 15) : sc; // (18 lanes: two DPP rows)
        const u64 hit = ballot(lane < n && cum >= 32u);
        const u32 count = hit ? (u32)__builtin_ctzll(hit) + 1u : n;
        const u32 last_end = rdl(pos + mylen, count - 1u);
        if (last_end <= 60u) { // (every position that mattered is a real lane: nothing was clamped)
            sum = rdl(cum, count - 1u);
            const u64 nzm = ballot(lane < count && myval != 0u);
            nonzero = (u32)__builtin_popcountll(nzm);
            // transmission index i -> symbol: 1,2,3,4,0,5,17,6,16,7,8,...,15; its inverse for lane sy
            const u32 inv = lane == 0u ? 4u : lane <= 4u ? lane - 1u : lane == 5u ? 5u : lane == 6u ? 7u : lane == 16u ? 8u : lane == 17u ? 6u : lane + 2u;
            const u32 k = inv - hskip; // (wraps for inv < hskip: then >= count)
            const u32 got = (u32)__shfl((int)myval, (int)(k & 63u));
            clv = (lane < 18u && k < count) ? got : 0u;
            if (nzm) {
                const u32 kl = 63u - (u32)__builtin_clzll(nzm), il = hskip + kl;
                single = il < 4u ? il + 1u : il == 4u ? 0u : il == 5u ? 5u : il == 6u ? 17u : il == 7u ? 6u : il == 8u ? 16u : il - 2u;
            }
            hb_skip(d, last_end > 32u ? 32u : last_end);
            if (last_end > 32u) hb_skip(d, last_end - 32u);
            if (sum > 32u) return ST_CODE_LENGTHS_CHECKSUM;
            parallel_done = true;
        }
    }
#endif
    // order of transmission: 1,2,3,4,0,5,17,6,16,7,8,...,15 (src/lib.rs:669)
    for (u32 i = hskip; i < 18u && !parallel_done; i++) {
        // fixed code (src/lib.rs:120-125), stream order: 00->0 01->3 10->4 110->2 1110->1 1111->5
        const u32 p = hb_peek(d) & 15u;
        u32 len, val;
        if ((p & 1u) == 0u) { len = 2; val = (p & 2u) ? 3u : 0u; }
        else if ((p & 2u) == 0u) { len = 2; val = 4u; }
        else if ((p & 4u) == 0u) { len = 3; val = 2u; }
        else { len = 4; val = (p & 8u) ? 5u : 1u; }
        hb_skip(d, len);
        if (val) {
            const u32 symi = i < 4u ? i + 1u : i == 4u ? 0u : i == 5u ? 5u : i == 6u ? 17u : i == 7u ? 6u : i == 8u ? 16u : i - 2u;
            clv = wrl(val, symi, clv);
            sum += 32u >> val;
            nonzero++;
            single = symi;
            if (sum == 32u) break;
            if (sum > 32u) return ST_CODE_LENGTHS_CHECKSUM;
        }
    }
    if (nonzero == 0u) return ST_NO_CODE_LENGTH;
    if (nonzero >= 2u && sum < 32u) return ST_CODE_LENGTHS_CHECKSUM;

    // 5-bit lookup table for the code-length code, one entry per lane (lanes 0..31): (symbol << 4) | len.  Canonical codes
    // lane-parallel: lane sy's code = first code of its length + the number of lower symbols of the same length.
    u32 cltab = 0;
    if (nonzero >= 2u) {
        const u32 l = lane < 18u ? clv : 0u;
        const u64 b0 = ballot((l & 1u) != 0u), b1 = ballot((l & 2u) != 0u), b2 = ballot((l & 4u) != 0u);
        u64 eq = (l & 1u) ? b0 : ~b0;
        eq &= (l & 2u) ? b1 : ~b1;
        eq &= (l & 4u) ? b2 : ~b2;
        const u32 rank = lanes_below(eq);
        u32 code = 0, mycode = 0;
        for (u32 k = 1; k <= 5u; k++) { // first code of each length (lengths 1..5: the fixed code's values)
            if (l == k) mycode = code + rank;
            const u64 mk = ((k & 1u) ? b0 : ~b0) & ((k & 2u) ? b1 : ~b1) & ((k & 4u) ? b2 : ~b2);
            code = (code + (u32)__builtin_popcountll(mk)) << 1;
        }
        const u32 rev = l ? __brev(mycode) >> (32u - l) : 0u;
        u64 todo = ballot(l != 0u);
        while (todo) { // every used symbol claims the table entries whose low `length` bits are its (bit-reversed) code
            const u32 sy = (u32)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const u32 ls = rdl(l, sy), rs = rdl(rev, sy);
            if ((lane & ((1u << ls) - 1u)) == rs) cltab = (sy << 4) | ls;
        }
    }

    lens_clear(d, s, alphabet);
    PT_ADD(7, pt);
    // The lengths themselves.  The 64 lengths of the chunk being decoded collect in the lanes of `cur` (v_writelane) and go
    // to s.lens with one byte store per lane when the chunk is left.
    u32 total = 0, i = 0, nz = 0;
    u32 last_symbol = 0xffu, last_repeat = 0, last_nz = 8;
    u32 cur = 0;
    bool dirty = false;
#define BRX_LENS_FLUSH(base_) do { if (dirty) s.lens[(base_) + lane] = (u8)cur; cur = 0u; dirty = false; } while (0)
    if (nonzero >= 2u) {
        // the loop below in hand-written assembly (brx_lens.S: same reader, same rules, ~22 instructions per length instead of
        // the ~60 the compiler makes of it); the one-symbol code-length code (no bits at all, Q5) stays with the C++ form
        u32 stat, nz_a, i_a, dirty_a, cur_a, vt0, vt1;
        // (readfirstlane on everything the statement takes in SGPRs: values LLVM's uniformity analysis gives up on would be an
        // "illegal VGPR to SGPR copy")
        u64 win = (u64)rfl((u32)d.win) | ((u64)rfl((u32)(d.win >> 32)) << 32);
        u32 nav = rfl(d.nav), ww = rfl(d.ww), cbase = rfl(d.cbase), cha = d.chunkA, chb = d.chunkB;
        asm volatile(
#ifdef BRX_SMALL
#include "_gen/brx_lens_asm_s.h"
#elif BRX_LEVEL == 0
#include "_gen/brx_lens_asm.h"
#elif BRX_LEVEL == 1
#include "_gen/brx_lens_asm_l1.h"
#elif BRX_LEVEL == 2
#include "_gen/brx_lens_asm_l2.h"
#elif BRX_LEVEL == 3
#include "_gen/brx_lens_asm_l3.h"
#else
#include "_gen/brx_lens_asm_l4.h"
#endif
            : "=s"(stat), "=s"(nz_a), "=s"(i_a), "=s"(dirty_a), "+s"(win), "+s"(nav), "+s"(ww), "+s"(cbase), "+v"(cha), "+v"(chb),
              "=&v"(cur_a), "=&v"(vt0), "=&v"(vt1)
            : "s"(alphabet), "s"(d.w_end), "s"(d.hb_lastw), "s"(d.hb_lastmask), "s"((u64)(uintptr_t)d.in_words), "v"(cltab), "v"(lane)
            : "memory", "vcc", "scc", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73",
              "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90");
        d.win = win; d.nav = nav; d.ww = ww; d.cbase = cbase; d.chunkA = cha; d.chunkB = chb;
        if (dirty_a) s.lens[((i_a - 1u) & ~63u) + lane] = (u8)cur_a;
        PT_ADD(8, pt);
        if (stat) return stat;
        if (nz_a < 2u) return ST_LESS_THAN_TWO_NONZERO;
        return ST_OK;
    }
    while (i < alphabet) {
        u32 sym;
        if (nonzero == 1u) {
            sym = single; // single-symbol code: zero bits (Q5)
        } else {
            const u32 e = rdl(cltab, hb_peek(d) & 31u);
            sym = e >> 4;
            hb_skip(d, e & 15u);
        }
        if (sym <= 15u) {
            cur = wrl(sym, i & 63u, cur);
            i++;
            last_symbol = sym;
            last_repeat = 0;
            if (sym) {
                dirty = true;
                nz++;
                last_nz = sym;
                total += 32768u >> sym;
                if (total == 32768u) break;
                if (total > 32768u) return ST_CODE_LENGTHS_CHECKSUM;
            }
            if ((i & 63u) == 0u) BRX_LENS_FLUSH(i - 64u);
        } else if (sym == 16u) {
            const u32 v = hb_bits(d, 2);
            u32 add, new_repeat;
            if (last_symbol == 16u && last_repeat) {
                new_repeat = 4u * (last_repeat - 2u) + v + 3u;
                add = new_repeat - last_repeat;
            } else {
                new_repeat = 3u + v;
                add = new_repeat;
            }
            if (i + add > alphabet) return ST_PARSE_COMPLEX_LENGTHS;
            nz += add;
            total += add * (32768u >> last_nz);
            for (u32 left = add; left != 0u;) { // the run, chunk by chunk
                const u32 a0 = i & 63u, k = left < 64u - a0 ? left : 64u - a0;
                cur = (lane - a0) < k ? last_nz : cur;
                dirty = true;
                i += k;
                left -= k;
                if ((i & 63u) == 0u) BRX_LENS_FLUSH(i - 64u);
            }
            if (total == 32768u) break;
            if (total > 32768u) return ST_CODE_LENGTHS_CHECKSUM;
            last_repeat = new_repeat;
            last_symbol = 16u;
        } else {
            const u32 v = hb_bits(d, 3);
            u32 ni;
            if (last_symbol == 17u && last_repeat) {
                const u32 new_repeat = 8u * (last_repeat - 2u) + v + 3u;
                ni = i + (new_repeat - last_repeat);
                last_repeat = new_repeat;
            } else {
                last_repeat = 3u + v;
                ni = i + last_repeat;
            }
            if (ni > alphabet) return ST_PARSE_COMPLEX_LENGTHS;
            if ((ni >> 6) != (i >> 6)) BRX_LENS_FLUSH(i & ~63u); // (the chunks in between stay zero: lens_clear)
            i = ni;
            last_symbol = 17u;
        }
    }
    BRX_LENS_FLUSH((i - 1u) & ~63u); // (dirty: position i - 1 was the last one written -- the loop may have left on a chunk's last symbol)
#undef BRX_LENS_FLUSH
    PT_ADD(8, pt);
    if (nz < 2u) return ST_LESS_THAN_TWO_NONZERO;
    return ST_OK;
}

// parse_prefix_code, src/lib.rs:877-889 = kind (:589-595) + simple (:597-665, Q8) or complex (:667-875, Q6, Q15)
FI u32 read_prefix_code(Dec &d, Lds &s, u32 alphabet, u32 &h, const bool wide = false) {
    PT_BEGIN(pt);
    const u32 kind = hb_bits(d, 2);
    if (kind == 1u) { // ---- simple
        const u32 bit_width = 32u - (u32)__builtin_clz(alphabet - 1u);
        const u32 nsym = hb_bits(d, 2) + 1u;
        u32 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        s0 = hb_bits(d, bit_width);
        if (s0 >= alphabet) return ST_INVALID_SYMBOL;
        if (nsym > 1u) {
            s1 = hb_bits(d, bit_width);
            if (s1 >= alphabet) return ST_INVALID_SYMBOL;
        }
        if (nsym > 2u) {
            s2 = hb_bits(d, bit_width);
            if (s2 >= alphabet) return ST_INVALID_SYMBOL;
        }
        if (nsym > 3u) {
            s3 = hb_bits(d, bit_width);
            if (s3 >= alphabet) return ST_INVALID_SYMBOL;
        }
        if (nsym > 1u && s0 == s1) return ST_INVALID_SYMBOL;
        if (nsym > 2u && (s0 == s2 || s1 == s2)) return ST_INVALID_SYMBOL;
        if (nsym > 3u && (s0 == s3 || s1 == s3 || s2 == s3)) return ST_INVALID_SYMBOL;
        u32 tree_select = 0;
        if (nsym == 4u) tree_select = hb_bits(d, 1);
        if (nsym == 1u) {
            h = build_single(d, s, s0);
            PT_ADD(6, pt);
            return ST_OK;
        }
        u32 l0, l1, l2, l3;
        if (nsym == 2u) { l0 = 1; l1 = 1; l2 = 0; l3 = 0; }
        else if (nsym == 3u) { l0 = 1; l1 = 2; l2 = 2; l3 = 0; }
        else if (!tree_select) { l0 = l1 = l2 = l3 = 2; }
        else { l0 = 1; l1 = 2; l2 = 3; l3 = 3; }
        h = build_simple(d, s, nsym, (l0 << 16) | s0, (l1 << 16) | s1, (l2 << 16) | s2, (l3 << 16) | s3, wide);
        PT_ADD(6, pt);
        return ST_OK;
    }
    const u32 rc = read_complex_lens(d, s, kind, alphabet);
    if (rc) return rc;
    PT_BEGIN(pb);
    h = build_code(d, s, alphabet, wide);
    PT_ADD(9, pb);
    return ST_OK;
}

// ---- output: LDS ring + aligned flush to HBM --------------------------------------------------------------
// Per-lane predication WITHOUT per-lane branches (a divergent branch anywhere in the command loop makes LLVM's
// uniformity analysis give up on the loop-carried decoder state and move all of it into VGPRs):
//   * LDS byte stores of switched-off lanes are steered to a per-lane trash byte;
//   * HBM stores / loads go through the stream's buffer resource, whose range check drops (reads as 0) any
//     access at offset >= num_records, so a switched-off lane simply uses offset 0xffffffff.
FI void ring_put(const Dec &d, Lds &s, bool on, u32 vpos, u32 byte) {
    u8 *p = on ? &s.ring[vpos & RMASK] : &s.trash[d.lane];
    *p = (u8)byte;
}
// ---- output stores: to the stream's slot in HBM (which is also its sliding window) and, when the caller's output buffer is
// pinned host memory, to the same offset of that buffer (BrxKernelArgs::out_mirror): the device-to-host copy rides on
// the decode as 1 KiB posted writes instead of following it.  Without a mirror the second resource has no records and the
// store is dropped by the range check -- no branch.
#define OUT_STORE128(q, off) do { const u32 off_ = (off); __builtin_amdgcn_raw_buffer_store_b128((q), d.out_rsrc, off_, 0, 0); \
                                  __builtin_amdgcn_raw_buffer_store_b128((q), d.out2_rsrc, off_, 0, 0); } while (0)
#define OUT_STORE8(b, off) do { const u32 off_ = (off); __builtin_amdgcn_raw_buffer_store_b8((u8)(b), d.out_rsrc, off_, 0, 0); \
                                __builtin_amdgcn_raw_buffer_store_b8((u8)(b), d.out2_rsrc, off_, 0, 0); } while (0)
// Flush skewed range [v0, v1) of the ring to HBM: full 16-B units as one 16-B store per lane, ragged head and
// tail (first / last block of a stream only) as byte stores.  v1 - v0 <= BRX_FLUSH_BLOCK + 15.
FI void flush_range(Dec &d, const Lds &s, u32 v0, u32 v1) {
    const u32 u0 = (v0 + 15u) >> 4, u1 = v1 >> 4; // full units [u0, u1)
    for (u32 ub = u0; ub < u1; ub += 64u) { // uniform trip count (1, rarely 2)
        u32 u = ub + d.lane;
        typedef u32 u32x4 __attribute__((ext_vector_type(4)));
        u32x4 q = *(const u32x4 *)&s.ring[(u * 16u) & RMASK];
        u32 off = u < u1 ? u * 16u - d.a : 0xffffffffu;
        OUT_STORE128(q, off);
    }
    if (v0 & 15u) { // ragged head: bytes [v0, min(u0*16, v1))
        u32 e = u0 * 16u < v1 ? u0 * 16u : v1;
        u32 v = v0 + d.lane;
        u32 b = s.ring[v & RMASK];
        OUT_STORE8(b, v < e ? v - d.a : 0xffffffffu);
    }
    if ((v1 & 15u) && u1 * 16u >= v0 && u1 >= u0) { // ragged tail: bytes [u1*16, v1)
        u32 v = u1 * 16u + d.lane;
        u32 b = s.ring[v & RMASK];
        OUT_STORE8(b, v < v1 ? v - d.a : 0xffffffffu);
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

// Source byte of one lane of a window copy: `back` = distance from the write cursor.  The ring holds the last
// BRX_RING_BYTES bytes, anything older is read back from the stream's own HBM output.  The choice between the
// two is made per wave whenever all lanes agree (almost always), so neither load sits behind a lane branch.
FI u32 copy_fetch(const Dec &d, const Lds &s, u32 back_max, u32 back_min, u32 back) {
    if (back_max <= BRX_RING_BYTES) return s.ring[(d.pos - back + d.a) & RMASK];
    if (back_min > BRX_RING_BYTES) return __builtin_amdgcn_raw_buffer_load_b8(d.out_rsrc, d.pos - back, 0, 0);
    u32 bn = s.ring[(d.pos - back + d.a) & RMASK];
    u32 bf = __builtin_amdgcn_raw_buffer_load_b8(d.out_rsrc, d.pos - back, 0, 0);
    return back <= BRX_RING_BYTES ? bn : bf;
}

// One KiB of a long copy: every lane moves 16 bytes.  Precondition: (pos + a) is 16-byte aligned, `de` >= 1024
// (so source and destination of this step do not overlap) and the source is entirely inside the ring
// (de <= BRX_RING_BYTES) or entirely older than it (de - 1023 > BRX_RING_BYTES; then it is final in HBM).
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
FI void bulk_copy_1k(Dec &d, Lds &s, u32 de) {
    const u32 vdst = d.pos + d.a + 16u * d.lane;
    u32x4 q;
    if (de <= BRX_RING_BYTES) {
        const u32 r = vdst - de; // skewed ring coordinate of this lane's 16 source bytes
        if ((de & 3u) == 0u) {
            q.x = *(const u32 *)&s.ring[r & RMASK];
            q.y = *(const u32 *)&s.ring[(r + 4u) & RMASK];
            q.z = *(const u32 *)&s.ring[(r + 8u) & RMASK];
            q.w = *(const u32 *)&s.ring[(r + 12u) & RMASK];
        } else {
            u32 w[4];
            _Pragma("unroll") for (u32 k = 0; k < 4u; k++) {
                u32 b0 = s.ring[(r + 4u * k) & RMASK], b1 = s.ring[(r + 4u * k + 1u) & RMASK];
                u32 b2 = s.ring[(r + 4u * k + 2u) & RMASK], b3 = s.ring[(r + 4u * k + 3u) & RMASK];
                w[k] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
            }
            q.x = w[0]; q.y = w[1]; q.z = w[2]; q.w = w[3];
        }
    } else {
        q = __builtin_amdgcn_raw_buffer_load_b128(d.out_rsrc, d.pos - de + 16u * d.lane, 0, 0);
    }
    *(u32x4 *)&s.ring[vdst & RMASK] = q;
    d.pos += 1024u;
}

// Window copy, reference copy_literals src/lib.rs:1491-1505: out[pos+i] = out[pos-dist + (i % dist)].
// Short copies go 64 bytes per step through the ring.  Long ones switch to 1 KiB steps (bulk_copy_1k) once the
// write cursor is 16-byte aligned and the effective distance is >= 1024: an overlapping (periodic) copy may use
// any multiple of its period as distance as soon as that much periodic history exists, so a period-1 or
// period-43 fill becomes a plain far-enough copy after its first ~1 KiB.
// A long overlapping copy is a periodic fill.  Once the ring holds one period P (a multiple of the distance AND of 16,
// <= 1 KiB) ending at a 16-byte aligned cursor, every further 1 KiB block is 64 aligned 16-byte units of that period:
// lane l reads its unit from the ring and stores it STRAIGHT to HBM -- no ring write, no flush read: one LDS read and
// one 16 B/lane buffer store per KiB and wave (configs 3/4 of BASELINE.json run at the HBM write rate this way).
// Afterwards the ring is re-seeded with the last 2 KiB of the output: the fill's last two blocks, again from the period in LDS.
#ifndef BRX_FILL_AUX
#define BRX_FILL_AUX 0 // cache policy bits of the fill's stores (A/B: 2 = nt)
#endif
FI void periodic_fill(Dec &d, Lds &s, u32 P, u32 nblocks) {
    flush_range(d, s, d.vfl, d.pos + d.a);                 // everything up to the cursor is in HBM now
    const u32 base = d.pos - P + d.a;                      // skewed ring coordinate of the period's first byte
    u32 o = (16u * d.lane) % P;                            // this lane's unit inside the period, for the first block
    const u32 step = 1024u % P;
    // The loop is the fill itself (64 blocks per backward65536 stream, 172 per quickfox_repeated): the next block's unit is
    // read while this one is stored (no LDS round trip inside an iteration), a period that divides 1 KiB (every power of two,
    // e.g. the period-1 fill) is read once, and the store to the host mirror is issued only where there is one.
    {
        u32x4 q = *(const u32x4 *)&s.ring[(base + o) & RMASK];
        u32 off = d.pos + 16u * d.lane;
        if (step == 0u) {
            if (d.mirror == nullptr) {
                for (u32 k = 0; k < nblocks; k++) { __builtin_amdgcn_raw_buffer_store_b128(q, d.out_rsrc, off, 0, BRX_FILL_AUX); off += 1024u; }
            } else {
                for (u32 k = 0; k < nblocks; k++) { OUT_STORE128(q, off); off += 1024u; }
            }
        } else {
            const bool both = d.mirror != nullptr;
            for (u32 k = 0; k < nblocks; k++) {
                o += step;
                o = o >= P ? o - P : o;
                const u32x4 qn = *(const u32x4 *)&s.ring[(base + o) & RMASK]; // (one read more than needed at the end)
                __builtin_amdgcn_raw_buffer_store_b128(q, d.out_rsrc, off, 0, BRX_FILL_AUX);
                if (both) __builtin_amdgcn_raw_buffer_store_b128(q, d.out2_rsrc, off, 0, 0);
                off += 1024u;
                q = qn;
            }
        }
        d.pos += nblocks << 10;
    }
    d.vfl = d.pos + d.a;
    // Re-seed the ring with the last 2 KiB of the output = the last two blocks of the fill: their units are still in LDS
    // (the period), so no store has to be waited for and nothing is read back from HBM.  Both blocks are read before the
    // first is written: the writes cover the whole ring, the period included.
    static_assert(BRX_RING_BYTES == 2048u, "periodic_fill re-seeds two 1 KiB blocks");
    if (nblocks >= 2u) {
        const u32 o1 = o >= step ? o - step : o + P - step;    // unit of block nblocks - 1
        const u32 o0 = o1 >= step ? o1 - step : o1 + P - step; // unit of block nblocks - 2
        const u32x4 q0 = *(const u32x4 *)&s.ring[(base + o0) & RMASK];
        const u32x4 q1 = *(const u32x4 *)&s.ring[(base + o1) & RMASK];
        const u32 off = d.pos - BRX_RING_BYTES + 16u * d.lane;
        *(u32x4 *)&s.ring[(off + d.a) & RMASK] = q0;
        *(u32x4 *)&s.ring[(off + 1024u + d.a) & RMASK] = q1;
        return;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): the stores are done before their bytes are read back
    for (u32 j = 0; j < BRX_RING_BYTES / 1024u; j++) {
        const u32 off = d.pos - BRX_RING_BYTES + 1024u * j + 16u * d.lane;
        const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(d.out_rsrc, off, 0, 0);
        *(u32x4 *)&s.ring[(off + d.a) & RMASK] = q;
    }
}

// A long copy whose source lies well behind the ring (distance >= 6 KiB: every source byte of a 4 KiB step is final in
// HBM once the ring is flushed): HBM -> registers -> HBM, 4 KiB per step with four 16 B/lane loads in flight per lane, no
// LDS traffic at all; both sides stream at 16 B per lane (the destination 16-byte aligned, the source as it falls).
// This is the LZ77 back-reference at memory speed (reference copy_literals, src/lib.rs:1483-1505); the ring is re-seeded
// from HBM afterwards, like after a periodic fill.
FI void direct_far_copy(Dec &d, Lds &s, u32 de, u32 nblocks) {
    flush_range(d, s, d.vfl, d.pos + d.a);
    for (u32 k = 0; k < nblocks; k++) {
        if (de < 16384u) __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the source was stored only a few steps ago
        const u32 src = d.pos - de + 16u * d.lane, dst = d.pos + 16u * d.lane;
        const u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(d.out_rsrc, src, 0, 0);
        const u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(d.out_rsrc, src + 1024u, 0, 0);
        const u32x4 q2 = __builtin_amdgcn_raw_buffer_load_b128(d.out_rsrc, src + 2048u, 0, 0);
        const u32x4 q3 = __builtin_amdgcn_raw_buffer_load_b128(d.out_rsrc, src + 3072u, 0, 0);
        OUT_STORE128(q0, dst);
        OUT_STORE128(q1, dst + 1024u);
        OUT_STORE128(q2, dst + 2048u);
        OUT_STORE128(q3, dst + 3072u);
        d.pos += 4096u;
    }
    d.vfl = d.pos + d.a;
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): the stores are done before their bytes are read back
    for (u32 j = 0; j < BRX_RING_BYTES / 1024u; j++) {
        const u32 off = d.pos - BRX_RING_BYTES + 1024u * j + 16u * d.lane;
        const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(d.out_rsrc, off, 0, 0);
        *(u32x4 *)&s.ring[(off + d.a) & RMASK] = q;
    }
}

FI void window_copy(Dec &d, Lds &s, u32 dist, u32 len, u32 &p1, u32 &p2) {
    u32 done = 0, de = dist;
    bool bulk_used = false;
    // period of an overlapping copy as a multiple of 16: dist * 16 / gcd(dist, 16)
    const u32 g16 = dist & (0u - dist) & 15u ? (dist & (0u - dist) & 15u) : 16u;
    const u32 P16 = dist * (16u / g16);
    while (done < len) {
        const u32 rem = len - done;
        if (dist < len && rem >= 8192u && P16 <= 1024u && done + dist >= P16 && ((d.pos + d.a) & 15u) == 0u) {
            const u32 nb = rem >> 10;
            periodic_fill(d, s, P16, nb);
            done += nb << 10;
            bulk_used = true;
            continue;
        }
        if (rem >= 16384u && de >= 4096u + BRX_RING_BYTES && ((d.pos + d.a) & 15u) == 0u) {
            const u32 nb = rem >> 12;
            direct_far_copy(d, s, de, nb);
            done += nb << 12;
            bulk_used = true;
            continue;
        }
        if (rem >= 2048u && de < 1024u) { // grow the distance to a multiple of the period >= 1024
            const u32 target = dist * ((1023u + dist) / dist);
            if (done + dist >= target) de = target;
        }
        if (rem >= 1024u && de >= 1024u && ((d.pos + d.a) & 15u) == 0u && (de <= BRX_RING_BYTES || de > BRX_RING_BYTES + 1023u)) {
            bulk_copy_1k(d, s, de);
            done += 1024u;
            bulk_used = true;
            maybe_flush(d, s);
            continue;
        }
        if (de < 64u && done + dist >= dist * ((63u + dist) / dist)) de = dist * ((63u + dist) / dist); // period-preserving distance >= 64
        if (rem >= 64u && rem < 1088u && de >= 64u && de <= BRX_RING_BYTES - 64u) {
            // the tail of a long copy (what a periodic fill or the 1 KiB steps leave over, < 1088 bytes), or a medium one: whole
            // 64-byte steps inside the ring, one LDS read and one LDS write each -- none of the per-step decisions below
            const u32 steps = rem >> 6;
            for (u32 k = 0; k < steps; k++) {
                const u32 v = d.pos + d.a + d.lane;
                s.ring[v & RMASK] = s.ring[(v - de) & RMASK];
                d.pos += 64u;
                maybe_flush(d, s);
            }
            done += steps << 6;
            bulk_used = true; // (p1 / p2 from the ring at the end)
            continue;
        }
        u32 n = rem < 64u ? rem : 64u;
        const u32 mis = (d.pos + d.a) & 15u;
        if (rem >= 1088u && mis) n = 16u - mis; // short step that aligns the cursor for the bulk path
        u32 lc = d.lane < n ? d.lane : n - 1u; // clamped lane: switched-off lanes redo the last byte
        u32 off = lc;
        if (de < n) off = lc % de; // overlapped copy shorter than a chunk: periodic source
        u32 back = de - off;       // distance of this lane's source byte from pos
        u32 off_max = de < n ? de - 1u : n - 1u;
        u32 b = copy_fetch(d, s, de, de - off_max, back);
        ring_put(d, s, d.lane < n, d.pos + d.lane + d.a, b);
        if (n >= 2u) { p1 = rdl(b, n - 1u); p2 = rdl(b, n - 2u); }
        else { p2 = p1; p1 = rdl(b, 0); }
        d.pos += n;
        done += n;
        maybe_flush(d, s);
        if (de < 64u && done + dist >= dist * ((63u + dist) / dist)) de = dist * ((63u + dist) / dist); // period-preserving distance >= 64
    }
    if (bulk_used) ctx_bytes(d, s, p1, p2);
}

// Static dictionary word + transform, reference src/lib.rs:1506-1540 and src/transformation/mod.rs:3-209.
// Returns a status; on ST_OK the word (wl bytes) sits in lanes 0..wl-1 of `wbyte`.
FI u32 dict_word(Dec &d, u32 copy_len, u32 word_id, u32 &wl, u32 &wbyte) {
    u32 nbits = K_NDBITS[copy_len];
    u32 index = word_id & ((1u << nbits) - 1u);
    u32 tid = word_id >> nbits;
    if (tid > 120u) return ST_INVALID_TRANSFORM_ID;
    const u8 *wp = d.t_dict + K_DOFFSET[copy_len] + index * copy_len;
    u32 w = (u32)wp[d.lane < copy_len ? d.lane : copy_len - 1u]; // clamped, lanes >= copy_len are never selected
    if (tid == 0u) { // the identity (Appendix B, transform 0: no prefix, no suffix): what encoders use most (alice29: all 721)
        wl = copy_len;
        wbyte = w;
        return ST_OK;
    }
    const BrxTransform *x = d.t_xforms + tid;
    u32 plen = rfl(x->plen), slen = rfl(x->slen), op = rfl(x->op);
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
    u32 pb = x->prefix[j & 7u], sb = x->suffix[(j - plen - mlen) & 7u]; // unconditional loads + bitwise selects
    u32 mp = 0u - (u32)(j < plen), mm = (0u - (u32)(j < plen + mlen)) & ~mp;
    wbyte = (pb & mp) | (mid & mm) | (sb & ~(mp | mm));
    return ST_OK;
}

// ---- block categories (reference: MetaBlock btype_x / blen_x, src/lib.rs:152-160) -----------------------
struct Cat {
    u32 nbl, btype, btype_prev, blen; // blen 0xffffffff = None (NBLTYPES == 1, Q12)
    u32 h_types, h_counts;
};

// parse_n_bltypes, src/lib.rs:501-525 (also NTREESL / NTREESD)
FI u32 read_n_bltypes(Dec &d) {
    if (!hb_bits(d, 1)) return 1u;
    const u32 k = hb_bits(d, 3);
    if (k == 0u) return 2u;
    return (1u << k) + 1u + hb_bits(d, k);
}
// Lookup in a code of the header (block types, block counts, context maps: 16-bit symbol entries) through the hb_* reader.
// `hv` = the tree's header words across the lanes (loaded once per tree by the caller: hd_tree).  An unassigned codeword
// of an incomplete code reads max_len + 1 bits and yields None (Q15), like decode_sym_as.
FI u32 hd_tree(const Dec &d, const Lds &s, u32 h) {
    return h < BRX_TM_WORDS ? tm_ld32<true>(d, s, h + (d.lane & 31u)) : tm_ld32<false>(d, s, h + (d.lane & 31u));
}
FI u32 hd_decode(Dec &d, const Lds &s, u32 h, u32 hv, u32 &sym) {
    const u32 h0 = rdl(hv, BRX_HDR_INFO);
    const u32 kind = h0 & 3u;
    if (kind == 0u) return LK_NONE;
    if (kind == 1u) {
        sym = h0 >> 16;
        return LK_OK;
    }
    const u32 v = __brev(hb_peek(d) & 0x7fffu) >> 17; // first stream bit = MSB of a 15-bit left-aligned code
    const u64 m = ballot(((v << 16) | 0xffffu) < hv) & 0xfffeull; // lanes 1 .. 15 carry limit << 16 | base
    if (m == 0ull) {
        hb_skip(d, ((h0 >> 8) & 0xffu) + 1u);
        return LK_NONE;
    }
    const u32 L = (u32)__builtin_ctzll(m);
    const u32 idx = ((v >> (15u - L)) + rdl(hv, L)) & 0xffffu;
    sym = h < BRX_TM_WORDS ? rfl(tm_ld16<true>(d, s, (h + BRX_HDR_WORDS) * 2u + idx)) : rfl(tm_ld16<false>(d, s, (h + BRX_HDR_WORDS) * 2u + idx));
    hb_skip(d, L);
    return LK_OK;
}
// first block count of a category (parse_first_block_count_* :989-1014 over parse_block_count :957-987): base and extra bits
// of the 26 block count codes in closed form (spec section 6)
FI u32 hd_block_count(Dec &d, const Lds &s, u32 h, u32 &blen) {
    u32 sym;
    if (hd_decode(d, s, h, hd_tree(d, s, h), sym) != LK_OK) return ST_EOF_FORMAT; // (no such codeword; cold_header turns a crossed end into ST_EOF)
    if (sym > 25u) return ST_INVALID_BLOCK_COUNT_CODE;
    u32 nb, base;
    if (sym < 16u) { const u32 g = sym >> 2; nb = 2u + g; base = 1u + 16u * ((1u << g) - 1u) + ((sym & 3u) << nb); }
    else if (sym < 18u) { nb = 6u; base = sym == 16u ? 241u : 305u; }
    else { nb = sym == 25u ? 24u : sym - 11u; base = 241u + (1u << (sym - 11u)); }
    blen = base + hb_bits(d, nb);
    return ST_OK;
}
// parse_block_count, src/lib.rs:957-987 (Ok(None) -> UnexpectedEOF, :977)
FI u32 read_block_count(Dec &d, const Lds &s, u32 h, u32 &blen) {
    u32 sym, e;
    const u32 lk = decode_sym(d, s, h, sym);
    if (lk != LK_OK) return lk == LK_EOF ? ST_EOF : ST_EOF_FORMAT;
    if (sym > 25u) return ST_INVALID_BLOCK_COUNT_CODE;
    u32 pk = rfl(K_BLEN[sym]);
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
// bytes [cm, cm+len), which the caller has zeroed: zero runs only move the cursor.  Values collect 64 at a time in the
// lanes of one register; the inverse move-to-front transform (:1164-1177) then runs over the stored map, again 64 entries
// per register, with the front 64 entries of its list in the lanes of m0 (a map rarely names a tree index >= 64).
FI u32 read_context_map_body(Dec &d, Lds &s, u32 h, u32 rlemax, u32 cm, u32 len) {
    PT_BEGIN(pt);
    const u32 lane = d.lane;
    const u32 hv = hd_tree(d, s, h);
    u32 pushed = 0, cur = 0;
    bool dirty = false;
#define BRX_CM_FLUSH(base_) do { if (dirty) { if (cm < TM_BYTES) { if ((base_) + lane < len) tm_st8<true>(d, s, cm + (base_) + lane, cur); } \
                                              else { if ((base_) + lane < len) tm_st8<false>(d, s, cm + (base_) + lane, cur); } } \
                                 cur = 0u; dirty = false; } while (0)
    while (pushed < len) {
        u32 code;
        const u32 lk = hd_decode(d, s, h, hv, code);
        if (lk != LK_OK) return ST_PARSE_CONTEXT_MAP;
        if (code > 0u && code <= rlemax) {
            const u32 repeat = (1u << code) + hb_bits(d, code);
            if (pushed + repeat > len) return ST_RUN_LENGTH_EXCEEDED;
            if (((pushed + repeat) >> 6) != (pushed >> 6)) BRX_CM_FLUSH(pushed & ~63u);
            pushed += repeat;
        } else {
            if (code) {
                cur = wrl(code - rlemax, pushed & 63u, cur);
                dirty = true;
            }
            pushed++;
            if ((pushed & 63u) == 0u) BRX_CM_FLUSH(pushed - 64u);
        }
    }
    if (pushed & 63u) BRX_CM_FLUSH(pushed & ~63u);
#undef BRX_CM_FLUSH
    const u32 imtf = hb_bits(d, 1);
    PT_ADD(10, pt);
    if (imtf) { // inverse_move_to_front_transform, src/lib.rs:1164-1177: the 256-entry list lives 4 per lane
        u32 m0 = lane, m1 = lane + 64u, m2 = lane + 128u, m3 = lane + 192u; // mtf[lane + 64*k]
        for (u32 base = 0; base < len; base += 64u) {
            const u32 cnt = len - base < 64u ? len - base : 64u;
            const u32 ba = cm + base + (lane < cnt ? lane : 0u);
            u32 vec = cm < TM_BYTES ? tm_ld8<true>(d, s, ba) : tm_ld8<false>(d, s, ba);
            // Only the NON-ZERO entries change the list (and most entries of a map are zeros: runs).  They are taken one by
            // one; a zero entry's value is the front of the list at its place = the value of the last non-zero entry in
            // front of it (or the front the chunk started with), fetched for all zero lanes at once behind the loop.
            const u32 front_in = rdl(m0, 0);
            const u64 nzmask = ballot(lane < cnt && vec != 0u);
            const u32 idxv = vec;
            for (u64 todo = nzmask; todo != 0ull; todo &= todo - 1ull) {
                const u32 k = (u32)__builtin_ctzll(todo);
                const u32 idx = rdl(idxv, k);
                if (idx < 64u) { // the front of the list: m0 alone moves (one DPP wave shift)
                    const u32 value = rdl(m0, idx);
                    const u32 up = (u32)__builtin_amdgcn_update_dpp((int)m0, (int)m0, 0x138, 0xf, 0xf, false); // wave_shr:1
                    m0 = lane == 0u ? value : (lane <= idx ? up : m0);
                    vec = wrl(value, k, vec);
                } else {
                    const u32 q = idx >> 6, l = idx & 63u;
                    const u32 value = q == 1u ? rdl(m1, l) : q == 2u ? rdl(m2, l) : rdl(m3, l);
                    vec = wrl(value, k, vec);
                    // shift mtf[0..idx) up by one, put value in front
                    const u32 c0 = rdl(m0, 63), c1 = rdl(m1, 63), c2 = rdl(m2, 63);
                    u32 s0 = (u32)__shfl_up((int)m0, 1), s1 = (u32)__shfl_up((int)m1, 1), s2 = (u32)__shfl_up((int)m2, 1),
                        s3 = (u32)__shfl_up((int)m3, 1);
                    if (lane == 0u) { s0 = value; s1 = c0; s2 = c1; s3 = c2; }
                    m0 = s0;
                    if (lane + 64u <= idx) m1 = s1;
                    if (lane + 128u <= idx) m2 = s2;
                    if (lane + 192u <= idx) m3 = s3;
                }
            }
            {
                const u64 below = nzmask & ((1ull << lane) - 1ull);
                const u32 src = below != 0ull ? 63u - (u32)__builtin_clzll(below) : 0u;
                const u32 fill = (u32)__shfl((int)vec, (int)src);
                vec = idxv != 0u ? vec : (below != 0ull ? fill : front_in);
            }
            if (cm < TM_BYTES) { if (lane < cnt) tm_st8<true>(d, s, cm + base + lane, vec); }
            else { if (lane < cnt) tm_st8<false>(d, s, cm + base + lane, vec); }
        }
        PT_ADD(11, pt);
    }
    return ST_OK;
}

FI u32 lut8(u32 vec, u32 b) { return (rdl(vec, b >> 2) >> ((b & 3u) * 8u)) & 0xffu; }

#ifndef BRX_SMALL
// ---- command loop ------------------------------------------------------------------------------------------
struct MB { // per-meta-block scalars the command loop needs
    u32 mlen, npostfix, ndirect, cmode_w, cml, cmd, hl, hi, hd, ntl, ntd;
};

// Meta-block header (reference states NBltypesL .. PrefixCodesDistances, src/lib.rs:1745-2002), out of line.
// Input: decoder state in Lds::st.  Output: status; on ST_OK the header results sit in Lds::mbw and the
// advanced input cursor / table-memory tops in Lds::st.
FI u32 header_body(Dec &d, Lds &s) {
    Cat L, I, D, cur;
    u32 rc;
    d.lds_top = 0;
    d.scr_top = 0;
    // ---- header: ONE loop whose tail reads "the next prefix code"; the head consumes the code read by the
    // previous iteration.  Order of fields = order of the reference's states NBltypesL .. PrefixCodesDistances.
    enum { S_CAT_N, S_CAT_TYPES, S_CAT_COUNTS, S_MISC, S_CM, S_CM_BODY, S_NTD, S_CODES_INIT, S_CODE };
    u32 step = S_CAT_N, c = 0, h = 0;
    u32 npostfix = 0, ndirect = 0, cmode_w = 0, ntl = 1, ntd = 1, cml = 0, cmd = 0, dalpha = 0;
    u32 cm = 0, cm_len = 0, rlemax = 0, which = 0, save_lds = 0, save_scr = 0, save_need = 0;
    u32 ht = 0, total = 0, idx = 0;
    L.nbl = I.nbl = D.nbl = 1; L.btype = I.btype = D.btype = 0; L.btype_prev = I.btype_prev = D.btype_prev = 1;
    L.blen = I.blen = D.blen = 0xffffffffu; L.h_types = I.h_types = D.h_types = 0; L.h_counts = I.h_counts = D.h_counts = 0;
    cur = L;
    for (;;) {
        u32 alphabet = 0;
        if (step == S_CAT_N) { // parse_n_bltypes_{l,i,d} :527-546 and what follows each (:1745-1885)
            cur.btype = 0; cur.btype_prev = 1; cur.blen = 0xffffffffu; cur.h_types = 0; cur.h_counts = 0;
            cur.nbl = read_n_bltypes(d);
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
            if ((rc = hd_block_count(d, s, h, cur.blen))) return rc; // parse_first_block_count_* :989-1014
            if (c == 0u) L = cur; else if (c == 1u) I = cur; else D = cur;
            c++;
            step = c < 3u ? S_CAT_N : S_MISC;
            continue;
        } else if (step == S_MISC) {
            npostfix = hb_bits(d, 2);            // parse_n_postfix :548
            ndirect = hb_bits(d, 4) << npostfix; // parse_n_direct :555
            dalpha = 16u + ndirect + (48u << npostfix);
            cmode_w = tm_alloc(d, (L.nbl + 3u) >> 2); // context modes, 2 bits per literal block type :562
            for (u32 base = 0; base < L.nbl; base += 64u) { // (64 modes collect in the lanes of one register)
                const u32 cnt = L.nbl - base < 64u ? L.nbl - base : 64u;
                u32 vec = 0;
                for (u32 i = 0; i < cnt; i++) vec = wrl(hb_bits(d, 2), i, vec);
                if (cmode_w < BRX_TM_WORDS) { if (d.lane < cnt) tm_st8<true>(d, s, cmode_w * 4u + base + d.lane, vec); }
                else { if (d.lane < cnt) tm_st8<false>(d, s, cmode_w * 4u + base + d.lane, vec); }
            }
            ntl = read_n_bltypes(d); // parse_n_trees_l :575
            cml = tm_alloc(d, 16u * L.nbl) * 4u; // 64 bytes per block type, zero = tree 0
            tm_zero_words(d, s, cml >> 2, 16u * L.nbl);
            if (ntl >= 2u) {
                cm = cml; cm_len = 64u * L.nbl; which = 0;
                step = S_CM;
                continue;
            } else {
                step = S_NTD;
                continue;
            }
        } else if (step == S_NTD) {
            ntd = read_n_bltypes(d); // parse_n_trees_d :582
            cmd = tm_alloc(d, D.nbl) * 4u; // 4 bytes per block type
            tm_zero_words(d, s, cmd >> 2, D.nbl);
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
            tm_set32(d, s, ht + idx, h);
            idx++;
            if (idx == total) break;
            // :1016 literals (256), :1034 insert&copy (704), :1052 distances (16 + NDIRECT + 48<<NPOSTFIX)
            alphabet = idx < ntl ? 256u : idx < ntl + I.nbl ? 704u : dalpha;
        }
        else if (step == S_CM) {
            // parse_context_map :1070-1144: RLEMAX, then a prefix code over rlemax+ntrees symbols, then the map
            rlemax = 0;
            if (hb_bits(d, 1)) rlemax = hb_bits(d, 4) + 1u;
            save_lds = d.lds_top; // the map's code is dead once the map is read
            save_scr = d.scr_top;
            save_need = d.need_cur;
            alphabet = rlemax + (which ? ntd : ntl);
            step = S_CM_BODY;
        } else if (step == S_CM_BODY) {
            if ((rc = read_context_map_body(d, s, h, rlemax, cm, cm_len))) return rc;
            d.lds_top = save_lds;
            d.scr_top = save_scr;
            d.need_cur = save_need;
            step = which ? S_CODES_INIT : S_NTD;
            continue;
        }
        // distance codes (the last ntd codes of S_CODE) keep 32-bit symbol entries: room for their payload form
        if ((rc = read_prefix_code(d, s, alphabet, h, step == S_CODE && idx >= ntl + I.nbl))) return rc;
    }
    const u32 hl = ht, hi = ht + ntl, hd = ht + ntl + I.nbl;
    if (d.lane == 0u) {
        u32 *w = s.mbw;
        w[0] = npostfix; w[1] = ndirect; w[2] = cmode_w; w[3] = cml; w[4] = cmd; w[5] = hl; w[6] = hi; w[7] = hd;
        w[8] = ntl; w[9] = ntd; w[10] = dalpha;
        w[MBW_ASM] = 0u; // symbol entries are plain symbols until prepare_fast_tables() says otherwise
        w[12] = L.nbl; w[13] = L.btype; w[14] = L.btype_prev; w[15] = L.blen; w[16] = L.h_types; w[17] = L.h_counts;
        w[18] = I.nbl; w[19] = I.btype; w[20] = I.btype_prev; w[21] = I.blen; w[22] = I.h_types; w[23] = I.h_counts;
        w[24] = D.nbl; w[25] = D.btype; w[26] = D.btype_prev; w[27] = D.blen; w[28] = D.h_types; w[29] = D.h_counts;
    }
    return ST_OK;
}
__device__ __noinline__ u32 cold_header() {
    Lds &s = g_lds;
    Dec d;
    PT_BEGIN(pl);
    dec_load_in(d, s);
    hb_begin(d);
    PT_ADD(13, pl);
    u32 rc = header_body(d, s);
    if (hb_over(d)) rc = ST_EOF; // a read crossed the end of the input: UnexpectedEOF came first (see hb_*)
    if (d.need_peak == 0xffffffffu) rc = ST_WATCHDOG; // no spill slab could be had (scratch_claim gave up): tables are not what they should be
    if (rc) {
        // (the slab this header may have claimed goes back with the stream: the dispatcher releases what st[20..21] names.  Until
        // round 4 an error behind the first spill lost it -- a slab less in the pool for the life of the context, and once a
        // launch's waves outnumbered what was left they waited for ever: tests/test_gpu_parity.py truncation sweep)
        put64(s, 20, (u64)(uintptr_t)d.scratch);
        return rc;
    }
    d.bitpos = hb_pos(d);
    dec_store_in(d, s);
    return ST_OK;
}

FI void mb_load(const Lds &s, MB &m, Cat &L, Cat &I, Cat &D) {
    const u32 *w = s.mbw;
    m.npostfix = rfl(w[0]); m.ndirect = rfl(w[1]); m.cmode_w = rfl(w[2]); m.cml = rfl(w[3]); m.cmd = rfl(w[4]);
    m.hl = rfl(w[5]); m.hi = rfl(w[6]); m.hd = rfl(w[7]); m.ntl = rfl(w[8]); m.ntd = rfl(w[9]);
    L.nbl = rfl(w[12]); L.btype = rfl(w[13]); L.btype_prev = rfl(w[14]); L.blen = rfl(w[15]); L.h_types = rfl(w[16]); L.h_counts = rfl(w[17]);
    I.nbl = rfl(w[18]); I.btype = rfl(w[19]); I.btype_prev = rfl(w[20]); I.blen = rfl(w[21]); I.h_types = rfl(w[22]); I.h_counts = rfl(w[23]);
    D.nbl = rfl(w[24]); D.btype = rfl(w[25]); D.btype_prev = rfl(w[26]); D.blen = rfl(w[27]); D.h_types = rfl(w[28]); D.h_counts = rfl(w[29]);
}

// The assembly command loop (brx_hot.S).  No operands: it reads and writes the parked state in LDS and returns the
// resume point it stopped at (0 = R0, 1 = R1, 2 = R2) in mbw[MBW_EXIT].
#define BRX_ASM_CLOBBERS \
        "memory", "vcc", "scc", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s29", "s30", "s31", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s24", "s25", "s26", "s27", "s28", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", \
          "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", \
          "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", \
          "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", \
          "s97", "s98", "s99", "s100", "s101", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", \
          "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", \
          "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", \
          "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "s11", "m0"
// Two builds of the same loop (brx_hot.S, "Two builds of this file"): the bit window in VGPRs -- for a full chip, where
// the CU's one scalar ALU is the busiest unit -- or in SGPRs -- for launches that leave the CUs mostly empty, where the
// shortest dependent chain wins.  BrxKernelArgs::loop_build picks one per launch.
__device__ __noinline__ u32 asm_commands() {
    asm volatile(
#if BRX_LEVEL == 0
#include "_gen/brx_hot_asm.h"
#elif BRX_LEVEL == 1
#include "_gen/brx_hot_asm_l1.h"
#elif BRX_LEVEL == 2
#include "_gen/brx_hot_asm_l2.h"
#elif BRX_LEVEL == 3
#include "_gen/brx_hot_asm_l3.h"
#else
#include "_gen/brx_hot_asm_l4.h"
#endif
        :
        :
        : BRX_ASM_CLOBBERS);
    return rfl(g_lds.mbw[MBW_EXIT]);
}
__device__ __noinline__ u32 asm_commands_sw() {
    asm volatile(
#if BRX_LEVEL == 0
#include "_gen/brx_hot_asm_sw.h"
#elif BRX_LEVEL == 1
#include "_gen/brx_hot_asm_sw_l1.h"
#elif BRX_LEVEL == 2
#include "_gen/brx_hot_asm_sw_l2.h"
#elif BRX_LEVEL == 3
#include "_gen/brx_hot_asm_sw_l3.h"
#else
#include "_gen/brx_hot_asm_sw_l4.h"
#endif
        :
        :
        : BRX_ASM_CLOBBERS);
    return rfl(g_lds.mbw[MBW_EXIT]);
}

// ---- the assembly loop's table forms -------------------------------------------------------------------------------
// A literal's share of the context id of the literals that follow it (reference parse_insert_literals context rules,
// src/lib.rs:1309-1338, src/lookuptable/mod.rs:1-56), laid out so that the loop gets the pre-scaled id with two masks:
//   id * 4 = (info(p1) & MA) | ((info(p2) & MB) << SB)
//   mode 0 LSB6:   info = (p & 63) << 2                MA 0xfc MB 0    SB 0
//   mode 1 MSB6:   info = (p >> 2) << 2                MA 0xfc MB 0    SB 0
//   mode 2 UTF8:   info = Lut0[p] << 2 | Lut1[p]       MA 0xfc MB 3    SB 2
//   mode 3 signed: info = Lut2[p] << 5 | Lut2[p] << 2  MA 0xe0 MB 0x1c SB 0
// (round 6: the three 256-byte LUTs sit across the lanes of the wave constants v_lut0 .. 2 -- lane l = bytes 4l .. 4l + 3 -- so a
// PER-LANE byte index is one shuffle instead of a load from global memory: the four steps over a literal tree's 256 entries waited
// ~1 us each for those loads, most of what HC_START cost a meta-block)
FI u32 lut_lane(u32 vec, u32 b) { return ((u32)__shfl((int)vec, (int)(b >> 2)) >> ((b & 3u) * 8u)) & 0xffu; }
FI u32 context_info(const Dec &d, u32 mode, u32 b) { // mode wave-uniform, b per lane (< 256); every lane must be active
    if (mode == 0u) return (b & 63u) << 2;
    if (mode == 1u) return (b >> 2) << 2;
    if (mode == 2u) return (lut_lane(d.v_lut0, b) << 2) | lut_lane(d.v_lut1, b);
    const u32 l2 = lut_lane(d.v_lut2, b);
    return (l2 << 5) | (l2 << 2);
}
// Payload form of one distance symbol (decode_distance, src/lib.rs:1412-1481).  A symbol whose base does not fit (only
// distances far beyond any window: 24 extra bits and more) becomes BRX_DIST_UNFIT | code: the assembly loop hands such
// a command to the C++ loop before consuming the symbol, the C++ loop decodes it from the code.
FI u32 distance_payload(u32 code, u32 npostfix, u32 ndirect) {
    if (code < 16u) return 0x80000000u | code;
    if (code < 16u + ndirect) return (code - 15u) << 5;
    const u32 x = code - ndirect - 16u;
    const u32 nbits = 1u + (x >> (npostfix + 1u));
    const u32 hcode = x >> npostfix, lcode = x & ((1u << npostfix) - 1u);
    const u64 offset = ((u64)(2u + (hcode & 1u)) << nbits) - 4ull;
    const u64 base = (offset << npostfix) + lcode + ndirect + 1u;
    if (nbits > 24u || base >= (1ull << 26)) return BRX_DIST_UNFIT | code;
    return nbits | ((u32)base << 5);
}
// Rewrite the symbol entries of a qualifying meta-block (all tables resident in LDS) into the assembly loop's forms.
// `uniform`: every literal block type has context mode `mode`, so the literal entries can carry the context info;
// otherwise they stay plain bytes and the loop looks the info up per literal (table rebuilt at every block switch).
// Always true since round 5 (the one shape that had no payload form -- a ONE-symbol distance tree whose symbol is not a last-distance
// code: its 16-bit slot in the info word cannot hold a base -- is materialised as a table, below).
FI bool prepare_fast_tables(const Dec &d, Lds &s, const MB &m, u32 n_iac, u32 mode, bool uniform) {
    for (u32 t = 0; t < m.ntd; t++) {
        const u32 h = rfl(s.tm[m.hd + t]);
        const u32 info = rfl(s.tm[h + BRX_HDR_INFO]);
        if ((info & 3u) == 1u && (info >> 16) >= 16u) {
            // A ONE-symbol distance code that names an EXPLICIT distance (every copy of the meta-block at one distance code: records,
            // low-entropy data; profiles/r05_fixture_rates.txt, e094_lowent).  Its payload does not fit the info word's 16 bits;
            // until round 5 such a meta-block stayed in the C++ loop.  Now the tree becomes a table the assembly loop's ordinary
            // lookup reads (as the one-symbol insert&copy code below): header[0] always matches at length 0, base -4 puts its two
            // candidate entries (32-bit) into header words 13 / 14 = the payload; kind bit 2 tells the assembly loop "general",
            // the C++ lookup still sees kind 1 and finds the marker 0xffff: the payload is in word 13 (decode_sym_as).
            const u32 pay = distance_payload(info >> 16, m.npostfix, m.ndirect);
            if (d.lane == 0u) {
                s.tm[h] = 0xffff0000u | ((0u - 4u) & 0xffffu);
                s.tm[h + 13u] = pay;
                s.tm[h + 14u] = pay;
                s.tm[h + BRX_HDR_INFO] = (info & 0xff00u) | 5u | 0xffff0000u;
            }
            continue;
        }
        if ((info & 3u) != 2u) continue; // (a one-symbol tree of a last-distance code keeps the plain code in its info word)
        const u32 nnz = info >> 16;
        for (u32 k = d.lane; k < nnz; k += 64u) s.tm[h + BRX_HDR_WORDS + k] = distance_payload(s.tm[h + BRX_HDR_WORDS + k], m.npostfix, m.ndirect);
    }
    for (u32 t = 0; t < n_iac; t++) {
        const u32 h = rfl(s.tm[m.hi + t]);
        const u32 info = rfl(s.tm[h + BRX_HDR_INFO]);
        if ((info & 3u) == 1u) {
            // A ONE-symbol insert&copy code (a meta-block whose commands are all alike -- e.g. one insert of 30 000 literals: low-entropy
            // data without repeats at quality 10 / 11 -- zero bits per symbol, Q5).  Round 5: it becomes a table the assembly loop's
            // ordinary lookup reads -- header[0] = "always matches" (limit all ones; lane 0 is the lowest lane: length 0, no bits) with
            // base -8, which puts its two candidate entries into header word 13 (unused: a one-symbol code has no lengths), both
            // = the symbol's record offset; the info word carries the same form for the C++ loop.  Before, such a meta-block ran in
            // the C++ loop alone: 2 700 cycles per literal (profiles/r05_fixture_rates.txt: e072_lowent 33 ms).
            const u32 x = ((info >> 16) << 4) & 0xffffu;
            if (d.lane == 0u) {
                s.tm[h] = 0xffff0000u | (0u - 8u & 0xffffu);
                s.tm[h + 13u] = x | (x << 16);
                s.tm[h + BRX_HDR_INFO] = (info & 0xffffu) | (x << 16);
            }
            continue;
        }
        const u32 nnz = info >> 16;
        u16 *sy = (u16 *)&s.tm[h + BRX_HDR_WORDS];
        for (u32 k = d.lane; k < nnz; k += 64u) sy[k] = (u16)(sy[k] << 4);
    }
    for (u32 t = 0; uniform && t < m.ntl; t++) {
        const u32 h = rfl(s.tm[m.hl + t]);
        const u32 info = rfl(s.tm[h + BRX_HDR_INFO]);
        if ((info & 3u) == 1u) { // one-symbol tree: the symbol lives in the info word
            const u32 sym = (info >> 16) & 0xffu;
            const u32 ci = context_info(d, mode, sym);
            if (d.lane == 0u) s.tm[h + BRX_HDR_INFO] = (info & 0xffffu) | ((sym | (ci << 8)) << 16);
            continue;
        }
        const u32 nnz = info >> 16;
        u16 *sy = (u16 *)&s.tm[h + BRX_HDR_WORDS];
        for (u32 k0 = 0; k0 < nnz; k0 += 64u) { // (uniform trip count: the shuffles of context_info want every lane)
            const u32 k = k0 + d.lane;
            const u32 sym = (k < nnz ? sy[k] : 0u) & 0xffu;
            const u32 ci = context_info(d, mode, sym);
            if (k < nnz) sy[k] = (u16)(sym | (ci << 8));
        }
    }
    return true;
}

// The table-memory command loop (reference states DataMetaBlockBegin .. CopyLiterals, src/lib.rs:2003-2141): every
// meta-block shape -- any NPOSTFIX / NDIRECT, any number of trees and block types, tables in LDS or in the HBM spill
// arena, exact end-of-input and capacity checks at every field.  Out of line.  It is also the safety net of the
// assembly loop (modes and resume points: HC_* above; a parked command travels in Lds::mbw).  Errors return at once
// (they are final).
FI u32 generic_body(Dec &d, Lds &s, const u32 mode_in) {
    const u32 mode = mode_in & 7u;
    const bool to_end = (mode_in & HC_TO_END) != 0u;
    MB m;
    Cat L, I, D;
    mb_load(s, m, L, I, D);
    m.mlen = rfl(s.st[ST_MLEN]);
    u32 rc;
    bool sw;
    u32 cmode = tm_u8(d, s, m.cmode_w * 4u + L.btype);
    u32 mb_left = m.mlen;
    u32 p1, p2;
    ctx_bytes(d, s, p1, p2);
    u32 insert_len = 0, copy_len = 0, implicit_zero = 0, distance = 0, dist_bad = 0;
    u32 budget = (mode == HC_WHOLE || mode == HC_RESUME_R1_WHOLE || to_end) ? 0xffffffffu : mode == HC_START ? 0u : 1u;
    const bool oneshot = mode >= HC_RESUME_R0 && mode <= HC_RESUME_R2 && !to_end;
    u32 phase2 = mode == HC_RESUME_R2 ? 1u : 0u;
    if (mode >= HC_RESUME_R0) {
        mb_left = rfl(s.mbw[MBW_MBLEFT]);
        insert_len = rfl(s.mbw[MBW_INS]); copy_len = rfl(s.mbw[MBW_CPY]); implicit_zero = rfl(s.mbw[MBW_IZ]);
        distance = rfl(s.mbw[MBW_DIST]); dist_bad = rfl(s.mbw[MBW_DISTBAD]);
    }
    if (mode == HC_START) {
        // Is this meta-block one for the assembly loop (preconditions in brx_hot.S)?
        const u32 total = m.ntl + I.nbl + m.ntd;
        u32 ok = (m.hl + total <= BRX_TM_WORDS && (u32)(uintptr_t)&g_lds == 0u && m.ntl <= 256u && m.ntd <= 256u &&
                  m.cml + 64u * L.nbl <= TM_BYTES && m.cmd + 4u * D.nbl <= TM_BYTES && m.cmode_w * 4u + L.nbl <= TM_BYTES &&
                  (u64)d.pos + m.mlen <= (u64)d.cap && d.bitend < (1ull << 31)) ? 1u : 0u;
        u32 why = ok ? 0u : ((m.ntl > 256u || m.ntd > 256u) ? 16u : (m.hl + total > BRX_TM_WORDS) ? 1u : 32u); // (bring-up statistics)
        // (up to 64 trees of a kind have their descriptors in the lanes of one register; beyond -- one piece of > 1 MiB from the reference encoder
        // has up to 256 literal trees -- the assembly loop gathers them from table memory: DESC_GATHER, round 5)
        for (u32 i = 0; i < total; i++) {
            const u32 h_ = ok ? tm_u32(d, s, m.hl + i) : 0u;
            if (h_ >= BRX_TM_WORDS - BRX_HDR_WORDS) { ok = 0u; why |= 2u; }
            else {
                const u32 kind_ = rfl(s.tm[h_ + BRX_HDR_INFO]) & 3u; // literal / distance trees may be one-symbol codes
                const bool iac_ = i >= m.ntl && i < m.ntl + I.nbl;
                if (kind_ != 2u && kind_ != 1u) { ok = 0u; why |= 4u; } // (general or one-symbol codes: prepare_fast_tables)
                (void)iac_;
                // a general code must be complete (the assembly lookup has no "no such codeword" exit, Q15):
                // the left-aligned upper bound of its longest codes is then exactly 2^15 (the high half of the header word)
                const u32 hvw_ = s.tm[h_ + (d.lane & 15u)];
                const bool full_ = ballot((hvw_ >> 16) == 0x8000u) != 0ull;
                if (kind_ == 2u && !full_) { ok = 0u; why |= 8u; }
            }
        }
        if ((rfl(s.st[ST_SPEC]) & 2u) != 0u) { ok = 0u; why |= 64u; } // (option command_loop = 6: as if no meta-block qualified -- the suite's way
                                                                       // to the path of meta-blocks the assembly loop cannot take)
        if (ok) { // one context mode for every literal block type (what encoders emit today): literal entries carry the info
            bool uniform = true;
            for (u32 i = 1; i < L.nbl; i++)
                if (tm_u8(d, s, m.cmode_w * 4u + i) != tm_u8(d, s, m.cmode_w * 4u)) uniform = false;
            if (!prepare_fast_tables(d, s, m, I.nbl, tm_u8(d, s, m.cmode_w * 4u), uniform)) { ok = 0u; why |= 128u; }
            else ok = uniform ? 1u : 3u; // bit 1: mixed context modes
        }
        s.mbw[MBW_ASM] = ok; // (where the loop is poisoned: Lds::mbw[MBW_WSAFE], set by the dispatcher before every call)
        if (!ok) s.pad[8] |= why;
    }
    const bool fast_tables = rfl(s.mbw[MBW_ASM]) != 0u; // symbol entries are in the assembly loop's forms

    // parse_insert_and_copy_length :1179-1208 + decode_insert_and_copy_length :1210-1224
#define G_DECODE_IAC()                                                                     \
    do {                                                                                   \
        if (++d.wd > d.wd_limit) return ST_WATCHDOG;                                       \
        if ((rc = cat_tick(d, s, I, sw))) return rc;                                       \
        u32 sym_;                                                                          \
        u32 lk_ = decode_sym(d, s, tm_u32(d, s, m.hi + I.btype), sym_);                    \
        if (lk_ == LK_NONE) return ST_PARSE_IAC;                                           \
        if (lk_ == LK_EOF) return ST_EOF;                                                  \
        if (fast_tables) sym_ >>= 4;                                                       \
        implicit_zero = sym_ < 128u ? 1u : 0u; /* :2012-2015 */                            \
        u32 cell_ = sym_ >> 6;                                                             \
        u32 ioff_ = (u32)((0x22120110000ull >> (4u * cell_)) & 15u) * 8u;                  \
        u32 coff_ = (u32)((0x21202101010ull >> (4u * cell_)) & 15u) * 8u;                  \
        u32 pki_ = rdl(d.v_ic, ioff_ + ((sym_ >> 3) & 7u));                                \
        u32 pkc_ = rdl(d.v_ic, 32u + coff_ + (sym_ & 7u));                                 \
        u32 e_;                                                                            \
        if (!in_bits(d, pki_ & 31u, e_)) return ST_EOF;                                    \
        insert_len = (pki_ >> 5) + e_;                                                     \
        if (!in_bits(d, pkc_ & 31u, e_)) return ST_EOF;                                    \
        copy_len = (pkc_ >> 5) + e_;                                                       \
    } while (0)
#define G_END_OF_COMMAND() budget = (oneshot && (d.vfl & (BRX_FLUSH_BLOCK - 1u)) != 0u) ? budget : budget - 1u

    if (mode <= HC_START || mode == HC_RESUME_R0) G_DECODE_IAC();
    while (mb_left != 0u && budget != 0u) {
        u32 max_allowed;
        if (phase2 == 0u) {
            if (insert_len > mb_left) return ST_EXCEEDED_EXPECTED_BYTES; // :2036 (Q4)
            if (!out_room(d, insert_len)) return ST_OUTPUT_TOO_SMALL;
            // ---- parse_insert_literals :1286-1365 (+ InsertLiterals state :2048-2081)
            for (u32 k = 0; k < insert_len; k++) {
                if ((rc = cat_tick(d, s, L, sw))) return rc;
                if (sw) cmode = tm_u8(d, s, m.cmode_w * 4u + L.btype);
                u32 cid;
                if (cmode == 3u) cid = (lut8(d.v_lut2, p1) << 3) | lut8(d.v_lut2, p2);
                else if (cmode == 2u) cid = lut8(d.v_lut0, p1) | lut8(d.v_lut1, p2);
                else if (cmode == 0u) cid = p1 & 0x3fu;
                else cid = p1 >> 2;
                u32 lit;
                const u32 ti = tm_u8(d, s, m.cml + L.btype * 64u + cid);
                const u32 lk = decode_sym(d, s, tm_u32(d, s, m.hl + ti), lit);
                if (lk == LK_NONE) return ST_PARSE_LITERALS;
                if (lk == LK_EOF) return ST_EOF;
                lit &= 0xffu; // (fast tables carry the context info in the high byte)
                ring_put(d, s, d.lane == 0u, d.pos + d.a, lit);
                d.pos++;
                p2 = p1;
                p1 = lit;
                if (((d.pos + d.a) & 63u) == 0u) maybe_flush(d, s);
            }
            if (insert_len) {
                mb_left -= insert_len;
                maybe_flush(d, s);
            }
            if (mb_left == 0u) continue; // :2069: the copy part of the last command is ignored
            // ---- parse_distance_code :1367-1410
            u32 dcode = 0;
            if (!implicit_zero) {
                if ((rc = cat_tick(d, s, D, sw))) return rc;
                const u32 cid = copy_len >= 5u ? 3u : copy_len - 2u;
                const u32 ti = tm_u8(d, s, m.cmd + D.btype * 4u + cid);
                const u32 lk = decode_sym_wide(d, s, tm_u32(d, s, m.hd + ti), dcode);
                if (lk == LK_NONE) return ST_PARSE_DISTANCE_CODE;
                if (lk == LK_EOF) return ST_EOF;
            }
            // ---- decode_distance :1412-1481
            if (fast_tables && (dcode & BRX_DIST_UNFIT) == BRX_DIST_UNFIT) dcode &= 0xffffu; // plain code: the arithmetic below
            else if (fast_tables && !implicit_zero && dcode >= 16u && (dcode & 0x80000000u) == 0u) {
                // payload form of a code >= 16: nbits | base << 5 (one-symbol trees: the plain code < 16, or the payload via decode_sym_as)
                u32 e;
                if (!in_bits(d, dcode & 31u, e)) return ST_EOF;
                distance = (dcode >> 5) + (e << m.npostfix);
                dcode = 0x10000u; // (only "not a last-distance code" matters below)
            } else if (fast_tables) dcode &= 0xffffu;
            if (dcode == 0x10000u) {
            } else if (dcode <= 3u) {
                distance = dcode == 0u ? d.dist0 : dcode == 1u ? d.dist1 : dcode == 2u ? d.dist2 : d.dist3;
            } else if (dcode <= 15u) {
                long long basev = dcode <= 9u ? (long long)d.dist0 : (long long)d.dist1;
                long long delta = dcode <= 9u ? (long long)((dcode - 2u) >> 1) : (long long)((dcode - 8u) >> 1);
                long long r = (dcode & 1u) ? basev + delta : basev - delta;
                if (r <= 0) return ST_NON_POSITIVE_DISTANCE;
                distance = (u32)r;
            } else if (dcode <= 15u + m.ndirect) {
                distance = dcode - 15u;
            } else {
                u32 e;
                const u32 x = dcode - m.ndirect - 16u;
                const u32 ndistbits = 1u + (x >> (m.npostfix + 1u));
                if (!in_bits(d, ndistbits, e)) return ST_EOF;
                const u32 hcode = x >> m.npostfix;
                const u32 lcode = x & ((1u << m.npostfix) - 1u);
                const u32 offset = ((2u + (hcode & 1u)) << ndistbits) - 4u;
                distance = ((offset + e) << m.npostfix) + lcode + m.ndirect + 1u;
            }
            max_allowed = d.pos < d.window ? d.pos : d.window;
            if (dcode > 0u && distance <= max_allowed) { // :1476-1478
                d.dist3 = d.dist2; d.dist2 = d.dist1; d.dist1 = d.dist0; d.dist0 = distance;
            }
        } else { // resumed at R2: the assembly loop decoded the distance (and updated the ring of last distances)
            max_allowed = d.pos < d.window ? d.pos : d.window;
            if (dist_bad) return ST_NON_POSITIVE_DISTANCE;
        }
        phase2 = 0u;
        // ---- copy_literals :1483-1542 (+ CopyLiterals state :2102-2141)
        if (distance <= max_allowed) {
            if (copy_len > mb_left) return ST_EXCEEDED_EXPECTED_BYTES; // :2105
            if (!out_room(d, copy_len)) return ST_OUTPUT_TOO_SMALL;
            mb_left -= copy_len;
            if (copy_len <= 64u && distance >= copy_len) {
                // the common case, software pipelined: issue the source read (LDS ring or the stream's own
                // HBM output), decode the NEXT command while the bytes are in flight, then land them.
                const u32 cl = copy_len; // the lookahead below overwrites insert_len / copy_len / implicit_zero
                const u32 lc = d.lane < cl ? d.lane : cl - 1u; // switched-off lanes redo the last byte
                const u32 b = copy_fetch(d, s, distance, distance - (cl - 1u), distance - lc);
                G_END_OF_COMMAND();
                if (mb_left != 0u) G_DECODE_IAC();
                ring_put(d, s, d.lane < cl, d.pos + d.lane + d.a, b);
                d.pos += cl;
                p1 = rdl(b, cl - 1u); // cl >= 2 always (copy length codes start at 2)
                p2 = rdl(b, cl - 2u);
                maybe_flush(d, s);
                continue;
            }
            window_copy(d, s, distance, copy_len, p1, p2);
        } else {
            if (copy_len < 4u || copy_len > 24u) return ST_INVALID_DICT_LENGTH;
            u32 wl, wb;
            if ((rc = dict_word(d, copy_len, distance - max_allowed - 1u, wl, wb))) return rc;
            if (wl > mb_left) return ST_EXCEEDED_EXPECTED_BYTES; // :2105 on the transformed length (Q4)
            if (!out_room(d, wl)) return ST_OUTPUT_TOO_SMALL;
            ring_put(d, s, d.lane < wl, d.pos + d.lane + d.a, wb);
            d.pos += wl;
            mb_left -= wl;
            if (wl >= 2u) { p1 = rdl(wb, wl - 1u); p2 = rdl(wb, wl - 2u); }
            else if (wl == 1u) { p2 = p1; p1 = rdl(wb, 0); }
            maybe_flush(d, s);
        }
        G_END_OF_COMMAND();
        if (mb_left != 0u) G_DECODE_IAC(); // :2128 otherwise
    }
#undef G_DECODE_IAC
#undef G_END_OF_COMMAND
    rc = ST_OK;
    if (mb_left != 0u) { // budget ran out: park at R1
        s.mbw[MBW_MBLEFT] = mb_left; s.mbw[MBW_INS] = insert_len; s.mbw[MBW_CPY] = copy_len; s.mbw[MBW_IZ] = implicit_zero;
        s.mbw[13] = L.btype; s.mbw[14] = L.btype_prev; s.mbw[15] = L.blen;
        s.mbw[19] = I.btype; s.mbw[20] = I.btype_prev; s.mbw[21] = I.blen;
        s.mbw[25] = D.btype; s.mbw[26] = D.btype_prev; s.mbw[27] = D.blen;
        rc = HC_CONTINUE;
    }
    return rc;
}
#endif // !BRX_SMALL
// per-lane constant vectors of the C++ command loop, loaded once per wave by the kernel (wave_consts) and passed by value
struct WaveConsts { u32 v_ic, v_lut0, v_lut1, v_lut2; };
FI WaveConsts wave_consts(const u32 *t_lut) {
    const u32 lane = threadIdx.x;
    WaveConsts c;
    const u32 ki = K_INS[lane < 24u ? lane : 23u], kc = K_COPY[(lane - 32u) < 24u ? lane - 32u : 23u];
    const u32 mi = 0u - (u32)(lane < 24u), mc = 0u - (u32)((lane - 32u) < 24u); // bitwise selects: branch-free
    c.v_ic = (ki & mi) | (kc & mc);
    c.v_lut0 = t_lut[lane]; c.v_lut1 = t_lut[64u + lane]; c.v_lut2 = t_lut[128u + lane];
    return c;
}
#ifndef BRX_SMALL
__device__ __noinline__ u32 generic_commands(u32 mode_in, u32 v_ic, u32 v_lut0, u32 v_lut1, u32 v_lut2) {
    Lds &s = g_lds;
    Dec d;
    dec_load(d, s);
    d.v_ic = v_ic; d.v_lut0 = v_lut0; d.v_lut1 = v_lut1; d.v_lut2 = v_lut2;
    const u32 rc = generic_body(d, s, rfl(mode_in)); // errors return from the middle: park the state here
    dec_store(d, s);
    return rc;
}

// Stream framing: reference decompress() states StreamBegin .. IsUncompressed / MLenLiterals and MetaBlockEnd ..
// StreamEnd (src/lib.rs:1550-1744, 2142-2167).  Runs until the stream ends (returns its final status) or a
// compressed meta-block starts (returns SEG_NEED_HEADER with MLEN / ISLAST parked in Lds::st).
__device__ __noinline__ u32 seg_frame() {
    Lds &s = g_lds;
    Dec d;
    dec_load(d, s);
    u32 v, b;
    u32 rc = ST_OK;
    bool finished = false;
    if (rfl(s.st[ST_STARTED]) == 0u) {
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
                if (v == 1u) return ST_EOF_FORMAT;
                wbits = v == 0u ? 17u : 8u + v;
            }
        }
        d.window = (1u << wbits) - 16u;
        if (d.lane == 0u) s.st[ST_STARTED] = 1u;
    } else if (rfl(s.st[ST_ISLAST]) != 0u) {
        finished = true; // back from the command loop of the last meta-block: MetaBlockEnd :2146-2153
    }
    const u64 pause_at = get64(s, ST_PAUSE_AT), in_low = get64(s, ST_IN_LOW);
    u64 mb_start = ~0ull;
    while (!finished) {
        if (++d.wd > d.wd_limit) { rc = ST_WATCHDOG; break; }
        // (resumable mode: uncompressed and metadata blocks are handled right here, one after the other -- a stream made of them
        // alone would run through all the resident input and out of the output window without ever reaching a pause point)
        if ((u64)d.pos >= pause_at || d.bitpos >= in_low) {
            dec_store(d, s);
            return SEG_PAUSED;
        }
        mb_start = d.bitpos;
        u32 is_last;
        if (!in_bits(d, 1, is_last)) { rc = ST_EOF; break; } // parse_is_last :420
        if (is_last) {
            if (!in_bits(d, 1, b)) { rc = ST_EOF; break; } // parse_is_last_empty :427
            if (b) break;
        }
        if (!in_bits(d, 2, v)) { rc = ST_EOF; break; } // parse_m_nibbles :434
        u32 mnibbles = v == 3u ? 0u : v + 4u;
        if (mnibbles == 0u) { // metadata block (accepted with ISLAST too, Q9), :1617-1683
            if (!in_bits(d, 1, b)) { rc = ST_EOF; break; }
            if (b) { rc = ST_NON_ZERO_RESERVED_BIT; break; }
            u32 mskipbytes;
            if (!in_bits(d, 2, mskipbytes)) { rc = ST_EOF; break; }
            if (mskipbytes == 0u) {
                if (in_byte_tail(d)) { rc = ST_NON_ZERO_FILL_BIT; break; }
            } else {
                u32 skip = 0, last = 0; // parse_m_skip_len :449-467: byte << i (Q2); errors -> EOF (Q10)
                bool bad = false;
                for (u32 i = 0; i < mskipbytes; i++) {
                    if (!in_bits(d, 8, last)) { bad = true; break; }
                    skip |= last << i;
                }
                if (bad) { rc = ST_EOF; break; }
                if (mskipbytes > 1u && last == 0u) { rc = ST_EOF_FORMAT; break; } // (Q10: whatever follows)
                skip += 1u;
                if (in_byte_tail(d)) { rc = ST_NON_ZERO_FILL_BIT; break; }
                if (in_remaining(d) < 8ull * skip) { rc = ST_EOF; break; }
                in_seek(d, d.bitpos + 8ull * skip);
            }
        } else {
            if (!in_bits(d, 4u * mnibbles, v)) { rc = ST_EOF; break; } // parse_m_len :469-483
            if (mnibbles > 4u && (v >> ((mnibbles - 1u) * 4u)) == 0u) { rc = ST_NON_ZERO_TRAILER_NIBBLE; break; }
            u32 mlen = v + 1u;
            u32 uncompressed = 0;
            if (!is_last && !in_bits(d, 1, uncompressed)) { rc = ST_EOF; break; } // :1689-1699
            if (uncompressed) { // :1701-1734
                if (in_byte_tail(d)) { rc = ST_NON_ZERO_FILL_BIT; break; }
                if (in_remaining(d) < 8ull * mlen) { rc = ST_EOF; break; }
                if (!out_room(d, mlen)) { rc = ST_OUTPUT_TOO_SMALL; break; }
                const u8 *src = (const u8 *)d.in_words + (d.bitpos >> 3);
                u32 done = 0;
                // Round 5: an uncompressed meta-block is a memcpy (incompressible payloads are stored this way by every encoder): up to
                // the next 16-byte boundary of the output through the ring, then INPUT -> registers -> OUTPUT, 4 KiB per step with four
                // 16 B/lane loads in flight (the source as it falls, the destination aligned: direct_far_copy's shape), the ring
                // re-seeded from the block's own last 2 KiB of input; the rest 64 bytes per step as before.  4096 x 256 KiB
                // of random bytes: 537 GB/s -> see profiles/r05_raw_ab.txt.
                if (mlen >= 8192u) {
                    const u32 head = (16u - ((d.pos + d.a) & 15u)) & 15u;
                    if (head) {
                        if (d.lane < head) s.ring[(d.pos + d.lane + d.a) & RMASK] = src[d.lane];
                        d.pos += head;
                        done = head;
                    }
                    flush_range(d, s, d.vfl, d.pos + d.a); // everything up to the cursor is in HBM now (ragged head as bytes)
                    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)d.in_words, 0, d.w_end * 4u, 0x00020000);
                    u32 so = (u32)(d.bitpos >> 3) + done + 16u * d.lane; // byte offset of this lane's unit in the input
                    const u32 nsteps = (mlen - done) >> 12;
                    for (u32 k = 0; k < nsteps; k++) {
                        const u32 dst = d.pos + 16u * d.lane;
                        const u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, so, 0, 0);
                        const u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, so + 1024u, 0, 0);
                        const u32x4 q2 = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, so + 2048u, 0, 0);
                        const u32x4 q3 = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, so + 3072u, 0, 0);
                        OUT_STORE128(q0, dst);
                        OUT_STORE128(q1, dst + 1024u);
                        OUT_STORE128(q2, dst + 2048u);
                        OUT_STORE128(q3, dst + 3072u);
                        d.pos += 4096u;
                        so += 4096u;
                    }
                    done += nsteps << 12;
                    d.vfl = d.pos + d.a;
                    for (u32 j = 0; j < BRX_RING_BYTES / 1024u; j++) { // the ring = the last 2 KiB of the output = of what was just copied
                        const u32 back = BRX_RING_BYTES - 1024u * j - 16u * d.lane; // bytes in front of the cursor
                        const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (u32)(d.bitpos >> 3) + done - back, 0, 0);
                        *(u32x4 *)&s.ring[(d.pos + d.a - back) & RMASK] = q;
                    }
                }
                for (; done < mlen; done += 64u) {
                    u32 n = mlen - done < 64u ? mlen - done : 64u;
                    if (d.lane < n) s.ring[(d.pos + d.lane + d.a) & RMASK] = src[done + d.lane];
                    d.pos += n;
                    maybe_flush(d, s);
                }
                in_seek(d, d.bitpos + 8ull * mlen);
            } else { // compressed meta-block: the dispatcher runs cold_header + a command loop, then comes back
                if (d.lane == 0u) { s.st[ST_ISLAST] = is_last; s.st[ST_MLEN] = mlen; }
                dec_store(d, s);
                return SEG_NEED_HEADER;
            }
        }
        if (is_last) break; // MetaBlockEnd :2146-2153
    }
    if (rc == ST_EOF && in_low != ~0ull && mb_start != ~0ull) {
        // (resumable mode, the source has more input than is resident: what looks like the end of the stream inside this
        // meta-block's framing -- an uncompressed block longer than what is left, say -- is the end of the WINDOW.  Back to the
        // meta-block's first bit -- nothing of it has been written yet -- and pause there; if it still fails with all the input
        // the host can make resident, the host asks again with in_low = ~0 and gets the status)
        in_seek(d, mb_start);
        dec_store(d, s);
        return SEG_PAUSED;
    }
    if (rc == ST_OUTPUT_TOO_SMALL && pause_at != ~0ull && mb_start != ~0ull) {
        // (resumable mode: an uncompressed meta-block larger than the room behind the output window -- back to its first bit, pause; the
        // host reads how far it runs from Lds::st[22] (Dec::needed) through the record and makes that room)
        in_seek(d, mb_start);
        dec_store(d, s);
        return SEG_PAUSED;
    }
    if (rc == ST_OK) { // StreamEnd :2155-2167
        if (in_byte_tail(d)) rc = ST_NON_ZERO_TRAILER_BIT;
        else if (d.bitpos < d.bitend) rc = ST_EXPECTED_END_OF_STREAM;
    }
    dec_store(d, s);
    return rc;
}

// Drain the ring to HBM at the end of a stream.
__device__ __noinline__ void seg_finish() {
    Lds &s = g_lds;
    Dec d;
    dec_load(d, s);
    if (d.vfl < d.pos + d.a) flush_range(d, s, d.vfl, d.pos + d.a);
}

// The kernel itself is only a dispatcher: per stream it parks the initial state in LDS and then alternates
// between the out-of-line segments.  Keeping it this small is what lets every segment have its own register
// allocation (nothing but `a`, `sid` and &s is live across the calls).
//
// Five instances of this file (BRX_LEVEL, brx_device.h): the regular kernel (10 KiB of LDS per wave, 16 waves per CU),
// three wider ones (12.5 / 20 / 40 KiB: 12 / 8 / 4 waves per CU) and, since round 5, level 4 (150 KiB, one per CU; fed by level 3 only).  A stream whose meta-block tables spill the table memory
// would run that meta-block in the C++ loop against tables in HBM (lcet10.txt: 8 x slower than its neighbours); with a.defer
// set a kernel instead drops such a stream at the first spill and lists it for the next level, whose kernel -- launched
// right behind on the same HIP stream, no host round trip -- decodes the listed streams from their start.
// (Every instance is compiled for 4 waves per SIMD = at most 128 VGPRs, the wider ones too although their LDS allows fewer waves:
// in a mixed batch the instances run next to each other on one CU, and a wave of a kernel compiled for 3 / 2 / 1 waves per SIMD
// is given 136 / 176 / 264 registers -- accumulation registers the compiler reserves because it may -- so that a SIMD holds one
// of them plus TWO regular waves instead of three: 12 waves per CU instead of 15, profiles/r04_two_queues.txt.)
#if BRX_LEVEL == 0
#define BRX_KERNEL_NAME brx_decode_kernel
#define BRX_LAUNCH_NAME brx_launch_decode
#define BRX_WAVES_PER_SIMD 4
#elif BRX_LEVEL == 1
#define BRX_KERNEL_NAME brx_decode_kernel_l1
#define BRX_LAUNCH_NAME brx_launch_decode_l1
#define BRX_WAVES_PER_SIMD 4
#elif BRX_LEVEL == 2
#define BRX_KERNEL_NAME brx_decode_kernel_l2
#define BRX_LAUNCH_NAME brx_launch_decode_l2
#define BRX_WAVES_PER_SIMD 4
#elif BRX_LEVEL == 3
#define BRX_KERNEL_NAME brx_decode_kernel_l3
#define BRX_LAUNCH_NAME brx_launch_decode_l3
#define BRX_WAVES_PER_SIMD 4
#else
#define BRX_KERNEL_NAME brx_decode_kernel_l4
#define BRX_LAUNCH_NAME brx_launch_decode_l4
#define BRX_WAVES_PER_SIMD 4
#endif
// The late list a level appends to / reads: levels 0 .. 2 append to the first one (count in word 8, stream indices in region 3 of
// `defer`), which the level-3 catch-all reads; level 3 appends to the second one (word 19, the index inside the record), level 4's input.
#if BRX_LEVEL == 3
#define LATE_OUT_RECS a.handup2
#define LATE_OUT_CAP a.late2_cap
#define LATE_OUT_CNT 19
#else
#define LATE_OUT_RECS a.handup
#define LATE_OUT_CAP a.late_cap
#define LATE_OUT_CNT 8
#endif
#if BRX_LEVEL == 4
#define LATE_IN_RECS a.handup2
#else
#define LATE_IN_RECS a.handup
#endif
// ---- hand-up of a stream to a wider level (BrxKernelArgs::defer) -------------------------------------------------------
// Table memory of levels 0 .. 3 in words (brx_device.h): the level a meta-block needing `need` words is LISTED for (level 4 is
// reached from level 3 only, through the second late list).
FI u32 level_for(u32 need) {
    return need <= 1728u + 2560u / 4u ? 1u : need <= 1728u + 10240u / 4u ? 2u : 3u;
}
// Words of a state record (the late list): what the reference's Decompressor carries across a meta-block boundary
// (src/lib.rs:1572-1573 resets everything of the meta-block; output position, window, the last four distances stay) plus
// this decoder's cursor and the framing of the meta-block whose header comes next.
enum { HU_BITPOS = 0, HU_POS = 2, HU_WINDOW = 3, HU_DIST = 4, HU_WD = 8, HU_ISLAST = 10, HU_MLEN = 11, HU_SID = 12, HU_WORDS = 16 };
// Resume with state: the ring takes the last BRX_RING_BYTES of the stream's output back from HBM (the kernel that handed the
// stream up flushed everything; it ran in an earlier launch, so its stores are visible -- the speculative end of the stream loop
// calls it behind a fence for this wave's own stores).  Units beyond the output's ends read
// as zeros or stale bytes: no command reads them before it has written them.
__device__ __noinline__ void seg_resume() {
    Lds &s = g_lds;
    Dec d;
    dec_load(d, s);
    const u32 top = d.pos + d.a; // (ring slot of a byte = (its position + a) mod the ring size: 16-byte units of ring and slot coincide)
    const u32 vend = (top + 15u) & ~15u;
    for (u32 j = 0; j < BRX_RING_BYTES / 1024u; j++) {
        const u32 v = vend - BRX_RING_BYTES + 1024u * j + 16u * d.lane;
        const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(d.out_rsrc, v - d.a, 0, 0);
        *(u32x4 *)&s.ring[v & RMASK] = q;
    }
    // The 16-byte unit that holds `top` came with whatever lies BEHIND the stream's last byte in its upper part; those slots
    // belong to the OLDEST bytes of the window (positions top - 2048 ...): a copy from 2033 .. 2048 bytes back right after the
    // resume read them (found by the round-4 soak: tools/wide_fuzz.py 3 43, three wrong bytes, only with an output slot that is
    // not 16-byte aligned; tests/golden/regress_late/r04_wide43_1_90).
    const u32 k = top & 15u;
    if (k != 0u && d.lane >= k && d.lane < 16u) {
        const u32 v = vend - 16u + d.lane;
        s.ring[v & RMASK] = (u8)__builtin_amdgcn_raw_buffer_load_b8(d.out_rsrc, v - BRX_RING_BYTES - d.a, 0, 0);
    }
    // ... and the unit that holds the stream's FIRST byte, while the window still reaches back to it (fewer than a ring's length of
    // output so far): with an output slot that is not 16-byte aligned that unit starts IN FRONT of the slot, its 16-byte load is out of
    // range as a whole and comes back as zeros -- the stream's first 15 bytes at most, which a copy from ~pos bytes back then read as
    // zeros (round 6: found by tools/node_fuzz.py -- a truncated stream whose speculative end goes back to a checkpoint at output
    // byte ~1 300, in a slot at offset 1 / 5; the late resume of rounds 4 / 5 had the same hole for hand-ups within the first 2 KiB).
#ifndef BRX_NO_FIRST_UNIT_FIX  // (A/B: tools/gpu_first_unit_ab.sh shows the test that fails without it)
    if (d.a != 0u && vend <= BRX_RING_BYTES && d.lane >= d.a && d.lane < 16u && d.lane < top)
        s.ring[d.lane] = (u8)__builtin_amdgcn_raw_buffer_load_b8(d.out_rsrc, d.lane - d.a, 0, 0);
#endif
}
__global__ __launch_bounds__(BRX_WAVE, BRX_WAVES_PER_SIMD) void BRX_KERNEL_NAME(BrxKernelArgs a) {
    Lds &s = g_lds;
    const u32 lane = threadIdx.x;
    if (a.debug_stop == 1u) return;
    const WaveConsts wc = wave_consts((const u32 *)a.t.context_lut);
#define generic_commands(m_) generic_commands((m_), wc.v_ic, wc.v_lut0, wc.v_lut1, wc.v_lut2)
    u32 *const counter = a.work_counter + a.counter_idx;
#if BRX_LEVEL > 0
    // (plan B: the host launches the next, narrower kernel only once every workgroup of the wider ones is resident -- workgroups
    // of 10 KiB that arrive first take three or four to a 40-KiB part of a CU's LDS and leave no room for a 20-KiB one there until
    // they all have left: brx_api.cpp launch())
    if (a.start_flag != nullptr && lane == 0u) {
        const u32 t = atomicAdd(a.work_counter + 14, 1u);
        if (t + 1u == a.start_total) {
            __threadfence_system();
            *a.start_flag = a.start_value;
        }
    }
    // A wider kernel decodes the lists of `list_mask` one behind the other: lists 0..2 hold streams to be decoded from their
    // start, list 3 (late) streams to be resumed from a state record.
    if (a.defer == nullptr) return;
    u32 cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0;
    if (a.list_mask & 1u) cnt0 = rfl(__builtin_nontemporal_load(&a.work_counter[5]));
    if (a.list_mask & 2u) cnt1 = rfl(__builtin_nontemporal_load(&a.work_counter[6]));
    if (a.list_mask & 4u) cnt2 = rfl(__builtin_nontemporal_load(&a.work_counter[7]));
    if (a.list_mask & 8u) cnt3 = rfl(__builtin_nontemporal_load(&a.work_counter[8]));
    cnt0 = cnt0 < a.defer_cap ? cnt0 : a.defer_cap; cnt1 = cnt1 < a.defer_cap ? cnt1 : a.defer_cap;
    cnt2 = cnt2 < a.defer_cap ? cnt2 : a.defer_cap; cnt3 = cnt3 < a.late_cap ? cnt3 : a.late_cap;
#if BRX_LEVEL == 4
    cnt0 = cnt1 = cnt2 = 0u; // (level 4 reads the second late list and nothing else)
    cnt3 = a.handup2 != nullptr ? rfl(__builtin_nontemporal_load(&a.work_counter[19])) : 0u;
    cnt3 = cnt3 < a.late2_cap ? cnt3 : a.late2_cap;
#endif
    const u32 n_streams = cnt0 + cnt1 + cnt2 + cnt3;
    if (n_streams == 0u) return;
    // few streams per CU: the sparse-launch build of the loop (levels 2 / 3 never have more than 8 / 4 per CU -- and their
    // streams are the many-tree ones, whose literals take that build's tree cache: 4096 x mapsdatazrh 61 -> 45 ms,
    // profiles/r05_slots_ab.txt)
    const bool sw_loop = a.loop_build != 0u || n_streams <= a.sw_threshold || (BRX_LEVEL >= 2 && a.sw_threshold != 0u);
#else
    // (behind the lean instance, BrxKernelArgs::s_list: queue slots [0, n) are this launch's own, the slots beyond are the
    // streams the lean kernel listed -- the large ones and the small ones it gave up on)
    const u32 n_listed = a.s_list != nullptr ? rfl(__builtin_nontemporal_load(&a.work_counter[10])) : 0u;
    const u32 n_streams = a.n + n_listed;
    if (n_streams == 0u) return;
    const bool sw_loop = a.loop_build != 0u;
    // Long jobs first, without a sort: when more streams are queued than workgroups run (the later ones start as the first ones
    // finish), the queue is walked four times -- for the streams of at least twice the mean compressed size, of at least the mean,
    // of at least half of it, then the rest.
    // A heterogeneous batch in the caller's order otherwise ends with its longest streams starting last (8192 streams of four
    // texts: 45.5 ms, longest first 33.4 ms; profiles/r04_order_ab.txt).  The mean: from the host (plan B: the pre-pass's
    // statistics of the streams that stay here), or -- device pointers, no queue order from the host -- from the lean kernel in
    // front, which saw every stream's size when it classified them (word 15 of the counter line: the sum in units of 64 B).
    u32 big = a.big_bytes;
    if (big == 0u && a.order == nullptr && a.cls == nullptr && a.n == 0u && n_listed != 0u && a.prepass == 0u) {
        const u64 sum = (u64)rfl(__builtin_nontemporal_load(&a.work_counter[15])) << 6;
        big = (u32)(sum / n_listed > 0xffffffffull ? 0xffffffffull : sum / n_listed);
    }
    const bool two_walk = big != 0u && n_streams > gridDim.x;
    // ... and only the walks that can have members are made: every ticket is an atomic on ONE address (~13 ns each, serialised),
    // and 16 384 equal streams walked four times are 49 152 tickets that find nothing (16 384 x monkey: 0.98 -> 1.43 ms).  Whoever
    // summed the sizes also kept the largest and the smallest (words 16 / 17, units of 64 B).
    u32 wlist = 0x3210u, nwalks = 4u;
    if (two_walk) {
        const u32 w16 = rfl(__builtin_nontemporal_load(&a.work_counter[16])), w17 = rfl(__builtin_nontemporal_load(&a.work_counter[17]));
        if (w17 != 0u) { // (0: nobody recorded them)
            const u64 max_end = ((u64)w16 + 1ull) << 6, min_len = (u64)(0xfffffu - w17) << 6; // max < max_end, min >= min_len
            const bool has0 = w16 >= 0xfffffu || max_end > 2ull * big, has2 = min_len < (u64)big, has3 = min_len < (u64)(big >> 1);
            wlist = 0u, nwalks = 0u;
            if (has0) wlist |= 0u << (4u * nwalks++);
            wlist |= 1u << (4u * nwalks++);
            if (has2) wlist |= 2u << (4u * nwalks++);
            if (has3) wlist |= 3u << (4u * nwalks++);
        }
    }
#endif
    // Work queue.  A wave's FIRST stream is its workgroup index, no atomic: 4096 waves adding to one address from eight
    // XCDs serialise at ~13 ns each (4096 EMPTY streams took 106 us that way, most of what config 3 took).  Later streams
    // come from the counter (queue slot = grid size + ticket).  Measured and dropped: a plain device-scope load of the
    // counter before the add (the contended line serves loads no faster), several slots per add after a tiny stream
    // (with 2 x grid streams a quarter of the waves then does the whole second round).
    bool first = true;
    for (;;) {
        u32 sid;
        u32 walk = 0u; // (index into the list of walks that are made)
        if (first) {
            first = false;
            // (workgroup i runs on XCD i % 8: with slot = i a batch whose streams repeat with a period of 2, 4 or 8 -- every fourth
            // stream the long one -- would put all its long streams on two XCDs.  The swizzle keeps every aligned block of 8 slots on
            // 8 different XCDs and varies which one takes which with the block's number; a ragged last block of 64 stays as it is.)
            sid = blockIdx.x;
            if ((sid | 63u) < gridDim.x) sid ^= (sid >> 3) & 7u;
        } else {
#if BRX_LEVEL == 0
            if (n_streams <= gridDim.x) break; // one stream per wave: nothing is queued
#else
            if (n_streams <= gridDim.x) break; // one stream per wave: nothing is queued
#endif
            // Every lane executes the atomic (only lane 0 adds): a lane-0-only branch here sits right behind the
            // lane-0-only status store that ends the previous iteration, and LLVM threads lanes 1..63 around both
            // across the back edge -- they then spin in their own loop and never meet lane 0 again.
            sid = gridDim.x + rdl(atomicAdd(counter, lane == 0u ? 1u : 0u), 0);
        }
#if BRX_LEVEL == 0
        if (two_walk && sid >= n_streams) { // the later walks over the queue: the smaller streams
            walk = sid / n_streams;
            sid -= walk * n_streams;
            if (walk >= nwalks) break;
        }
#endif
        if (sid >= n_streams) break;
#if BRX_LEVEL > 0
        u32 late_slot = 0xffffffffu; // >= 0: resume from state record `late_slot`
        if (sid < cnt0) sid = rfl(a.defer[sid]);
        else if (sid < cnt0 + cnt1) sid = rfl(a.defer[(size_t)a.defer_cap + (sid - cnt0)]);
        else if (sid < cnt0 + cnt1 + cnt2) sid = rfl(a.defer[2u * (size_t)a.defer_cap + (sid - cnt0 - cnt1)]);
        else {
            late_slot = sid - cnt0 - cnt1 - cnt2;
#if BRX_LEVEL == 4
            sid = rfl(__builtin_nontemporal_load(&a.handup2[(size_t)late_slot * HU_WORDS + HU_SID]));
#else
            sid = rfl(a.defer[3u * (size_t)a.defer_cap + late_slot]);
#endif
        }
#else
        if (sid >= a.n) sid = rfl(a.s_list[sid - a.n]);        // listed by the lean kernel
        else if (a.order != nullptr) sid = rfl(a.order[sid]); // the host path queues the longest streams first
        // plan B: the pre-pass has classified every stream of this queue; the ones of the wider levels are theirs
        if (a.cls != nullptr && a.prepass == 0u && rfl((u32)a.cls[sid]) != 0u) continue;
#endif
        const u64 i0 = a.in_off[sid], i1 = a.in_off[sid + 1u];
        const u64 o0 = a.out_off[sid], o1 = a.out_off[sid + 1u];
#if BRX_LEVEL == 0
        if (two_walk) { // walk 0: sizes >= 2 x mean, 1: [mean, 2 x mean), 2: [mean / 2, mean), 3: the rest
            const u64 len = i1 >= i0 ? i1 - i0 : 0ull;
            const u32 mine = len >= 2ull * big ? 0u : len >= (u64)big ? 1u : len >= (u64)(big >> 1) ? 2u : 3u;
            if (mine != ((wlist >> (4u * walk)) & 15u)) continue; // another walk's
        }
#endif
        {
            Dec d;
            d.lane = lane;
            const u8 *inp = a.in + i0;
            const u32 mis = (u32)((uintptr_t)inp & 3u);
            d.in_words = (const u32 *)(inp - mis);
            const u64 in_len = i1 >= i0 ? i1 - i0 : 0ull; // a decreasing offset table gives an empty stream, never a wild range
            d.w_end = (u32)((mis + in_len + 3u) >> 2);
            d.bitend = 8ull * (mis + in_len);
            d.bitpos = 8ull * mis;
            d.out = a.out + o0;
            d.mirror = a.out_mirror != nullptr ? a.out_mirror + o0 : nullptr;
            const u64 capacity = o1 >= o0 ? o1 - o0 : 0ull;
            d.cap = capacity > 0xffffff00ull ? 0xffffff00u : (u32)capacity;
            d.pos = 0;
            d.a = (u32)((uintptr_t)d.out & 15u);
            d.vfl = d.a;
            d.window = 0;
            d.dist0 = 4; d.dist1 = 11; d.dist2 = 15; d.dist3 = 16; // src/lib.rs:408
            d.needed = 0;
            d.wd = 0;
            d.wd_limit = 8ull * in_len + (u64)d.cap + 65536ull;
            d.lds_top = 0;
            d.scr_top = 0;
            d.scratch = nullptr;
            d.pool = a.pool;
            d.t_dict = a.t.dict;
            d.t_xforms = a.t.xforms;
            d.t_lut = (const u32 *)a.t.context_lut;
            dec_store(d, s);
            if (lane == 0u) {
                s.st[ST_STARTED] = 0u; s.st[ST_ISLAST] = 0u; s.st[ST_MLEN] = 0u;
                s.st[ST_IACTAB] = (u32)(uintptr_t)a.t.iac; s.st[ST_IACTAB + 1] = (u32)((u64)(uintptr_t)a.t.iac >> 32);
                s.st[ST_PAUSE_AT] = 0xffffffffu; s.st[ST_PAUSE_AT + 1] = 0xffffffffu;
                s.st[ST_IN_LOW] = 0xffffffffu; s.st[ST_IN_LOW + 1] = 0xffffffffu;
                s.st[ST_SPEC] = ((a.resume == nullptr && a.debug_stop == 0u) ? 1u : 0u) | (a.debug_stop == 6u ? 2u : 0u);
            }
            if (lane < 32u) s.pad[lane] = 0u;
        }
#if BRX_LEVEL == 0
        if (a.resume != nullptr) {
            // ---- resumable mode: one stream decoded in slices against a sliding output window (brx_api.cpp, streaming
            // Read facade).  Pauses only between the out-of-line segments, where the whole state sits in LDS.
            BrxResume *rec = a.resume + sid;
            enum { PH_FRAME = 0, PH_LOOP = 1, PH_HEADER = 2, PH_ASMEXIT = 3 };
            u32 phase = PH_FRAME;
            u32 st = 0;
            if (rfl(rec->state) == 1u) {
                u32 *dst = (u32 *)&s;
                for (u32 w = lane; w < BRX_LDS_BYTES / 4u; w += 64u) dst[w] = rec->lds[w];
                if (lane == 0u) { // the output window has moved: same bytes, new base (a multiple of 16 away) and capacity
                    const u64 op = (u64)(uintptr_t)(a.out + o0);
                    const u64 capacity = o1 >= o0 ? o1 - o0 : 0ull;
                    const u32 cap = capacity > 0xffffff00ull ? 0xffffff00u : (u32)capacity;
                    s.st[7] = (u32)op; s.st[8] = (u32)(op >> 32); s.st[9] = cap;
                    // ... and so may the input window have (BrxResume::in_slide): new base / end, the cursor relative to it; the
                    // loop watchdog counts per slice (its limit follows from what is resident)
                    const u8 *inp = a.in + i0;
                    const u32 mis = (u32)((uintptr_t)inp & 3u);
                    const u64 iw = (u64)(uintptr_t)(inp - mis), in_len = i1 >= i0 ? i1 - i0 : 0ull;
                    s.st[0] = (u32)iw; s.st[1] = (u32)(iw >> 32); s.st[2] = (u32)((mis + in_len + 3u) >> 2);
                    const u64 be = 8ull * (mis + in_len);
                    s.st[5] = (u32)be; s.st[6] = (u32)(be >> 32);
                    const u64 bp = ((u64)s.st[3] | ((u64)s.st[4] << 32)) - 8ull * rec->in_slide;
                    s.st[3] = (u32)bp; s.st[4] = (u32)(bp >> 32);
                    s.st[29] = 0u; s.st[30] = 0u;
                    const u64 wdl = 8ull * in_len + (u64)cap + 65536ull;
                    s.st[31] = (u32)wdl; s.st[32] = (u32)(wdl >> 32);
                }
                phase = rfl(rec->phase);
                st = (phase == PH_LOOP || phase == PH_ASMEXIT) ? HC_CONTINUE : 0u;
            }
            const u64 pause_at = rec->pause_at, in_low = rec->in_low;
            const bool no_mb_pause = rec->need_room == 1ull; // (on the way in: the host could not make room for a whole meta-block before)
            bool room_optional = false;
            if (lane == 0u) {
                s.st[ST_PAUSE_AT] = (u32)pause_at; s.st[ST_PAUSE_AT + 1] = (u32)(pause_at >> 32);
                s.st[ST_IN_LOW] = (u32)in_low; s.st[ST_IN_LOW + 1] = (u32)(in_low >> 32);
            }
            bool paused = false;
            u32 need_room = 0u; // an item that did not fit the output window's capacity was taken back: the position it runs to
            if (lane == 0u) s.st[22] = 0u;
            for (;;) {
                if (phase == PH_FRAME) {
                    if (((u64)rfl(s.st[10]) >= pause_at || get64(s, 3) >= in_low) && rfl(s.st[ST_STARTED]) != 0u && rfl(s.st[ST_ISLAST]) == 0u) { paused = true; break; }
                    st = seg_frame();
                    if (st == SEG_PAUSED) { paused = true; need_room = rfl(s.st[22]); break; }
                    if (st != SEG_NEED_HEADER) break;
                    phase = PH_HEADER;
                }
                // A segment that runs into the end of the RESIDENT input while the source has more (in_low != ~0) has not met the end
                // of the stream: the slice pauses IN FRONT of it and the next one, with more input resident, runs it again.  The
                // header stores its cursor only when it succeeds; a command of the C++ loop parks its state only when it succeeds too,
                // but leaves Lds::st as it was when it failed -- so st as it stood before the call (one word per lane of a
                // register) goes back, bar the flush cursor: what the failed call has flushed are good bytes, and they stay flushed.
#define BRX_ST_BACKUP() const u32 st_bak = s.st[lane < 48u ? lane : 0u]
#define BRX_ST_RESTORE() do { if (lane < 48u && lane != 12u && lane != 20u && lane != 21u) s.st[lane] = st_bak; } while (0) /* (a slab claimed meanwhile stays claimed) */
                // ... and the RING: a command of the C++ loop inserts its literals before its copy finds no room (or its next field no
                // input), and an insert of RING - 2 bytes or more has overwritten the slots of the bytes in front of the command -- the two
                // context bytes the re-run's first literals choose their tree by (a wrong tree with two or more literal
                // trees, silently).  What the failed call produced is flushed, so the last 2 KiB in front of the restored position come
                // back from the stream's own output, as for a late resume.  And the FLUSH CURSOR goes back to the position: what the failed
                // call flushed beyond it are good bytes, but the host slides / re-allocates the output window before the next slice and
                // keeps only what lies in front of the position (round 6, found by the test of the above: the first 4 KiB of such an
                // insert came back as zeros once the buffer had grown) -- the re-run flushes them again.
#define BRX_RING_BACK() do { seg_finish(); __threadfence(); if (lane == 0u) s.st[12] = s.st[10] + s.st[11]; seg_resume(); } while (0)
                if (phase == PH_HEADER) {
                    // Room for the WHOLE meta-block behind the window?  The assembly loop runs a meta-block only when it fits the capacity
                    // (it has no capacity test per command), so a large meta-block that started deep in the window used to run in the C++
                    // loop, one command per call -- ~3 MB/s instead of ~25 for exactly the streams a reader is for (big files in 16 MiB
                    // meta-blocks).  Now the slice pauses in FRONT of such a meta-block and says how far it runs: the host
                    // slides the window and, if need be, lets the buffer grow to window + meta-block (BrxResume::need_room, as for one
                    // oversized command); nothing has been parsed yet, nothing is taken back.
                    {
                        const u64 mb_end = (u64)rfl(s.st[10]) + rfl(s.st[ST_MLEN]);
                        if (pause_at != ~0ull && !no_mb_pause && mb_end > (u64)rfl(s.st[9]) && mb_end <= 0xffffff00ull) {
                            need_room = (u32)mb_end;
                            room_optional = true; // (bit 63 of BrxResume::need_room: the host may fail to make this room, the meta-block decodes anyway)
                            paused = true;
                            break;
                        }
                    }
                    BRX_ST_BACKUP();
                    st = cold_header();
                    if (st == ST_OK) st = generic_commands(HC_START);
                    if (st == ST_EOF && in_low != ~0ull) { BRX_ST_RESTORE(); paused = true; break; }
                    if (st != HC_CONTINUE && st != ST_OK) break;
                    phase = PH_LOOP;
                }
                while (st == HC_CONTINUE) {
                    u32 r;
                    if (phase == PH_ASMEXIT) { // resumed right behind an exit of the assembly loop: the command it handed back is due
                        r = rfl(s.mbw[MBW_EXIT]);
                        phase = PH_LOOP;
                    } else {
                        if ((u64)rfl(s.st[10]) >= pause_at || get64(s, 3) >= in_low) { paused = true; break; }
                        if (rfl(s.mbw[MBW_ASM]) == 0u) {
                            BRX_ST_BACKUP();
                            st = generic_commands(HC_RESUME_R1); // one command per call: a pause point after each
                            if (st == ST_EOF && in_low != ~0ull) { BRX_ST_RESTORE(); BRX_RING_BACK(); st = HC_CONTINUE; paused = true; break; }
                            if (st == ST_OUTPUT_TOO_SMALL) { need_room = rfl(s.st[22]); BRX_ST_RESTORE(); BRX_RING_BACK(); st = HC_CONTINUE; paused = true; break; }
                            continue;
                        }
                        if (lane == 0u) { const u32 we = (u32)(get64(s, 5) >> 5); s.mbw[MBW_WSAFE] = we > BRX_END_MARGIN ? we - BRX_END_MARGIN : 0u; }
                        r = sw_loop ? asm_commands_sw() : asm_commands();
                        // (the assembly loop knows no pause: it runs until something unusual comes up -- at the latest the last dwords
                        // of the RESIDENT input, and the command it hands back there may well straddle that end.  While the source
                        // has more the slice pauses HERE, in front of that command, instead of failing in it)
                        if (get64(s, 3) >= in_low) { phase = PH_ASMEXIT; paused = true; break; }
                    }
                    {
                        BRX_ST_BACKUP();
                        st = generic_commands(HC_RESUME_R0 + ((r & 3u) > 2u ? 1u : (r & 3u)));
                        if (st == ST_EOF && in_low != ~0ull) { BRX_ST_RESTORE(); BRX_RING_BACK(); st = HC_CONTINUE; phase = PH_ASMEXIT; paused = true; break; }
                        // (one command -- a long copy, a long insert -- that runs past the window's capacity: taken back like the one that
                        // ran out of input; the host makes room for it, BrxResume::need_room)
                        if (st == ST_OUTPUT_TOO_SMALL) { need_room = rfl(s.st[22]); BRX_ST_RESTORE(); BRX_RING_BACK(); st = HC_CONTINUE; phase = PH_ASMEXIT; paused = true; break; }
                    }
                }
#undef BRX_ST_BACKUP
#undef BRX_ST_RESTORE
#undef BRX_RING_BACK
                if (paused || st) break;
                phase = PH_FRAME;
            }
            seg_finish(); // everything produced so far is in HBM
            if (paused) {
                const u32 *src = (const u32 *)&s;
                for (u32 w = lane; w < BRX_LDS_BYTES / 4u; w += 64u) rec->lds[w] = src[w];
                if (lane == 0u) { rec->state = 1u; rec->phase = phase; rec->need_room = (u64)need_room | (room_optional ? 1ull << 63 : 0ull); }
                st = BRX_PAUSED;
            } else {
                const u32 *slab = (const u32 *)(uintptr_t)get64(s, 20);
                if (slab != nullptr) scratch_release(a.pool, slab);
                if (lane == 0u) rec->state = 2u;
            }
            if (lane == 0u) {
                a.status[sid] = (int)final_status(st);
                a.out_len[sid] = st == ST_OUTPUT_TOO_SMALL ? (u64)rfl(s.st[22]) : (u64)rfl(s.st[10]);
            }
            continue;
        }
#endif
#ifdef BRX_BRINGUP // (BRX_BRINGUP=1 at build time: per-stream statistics and LDS dumps; not in the shipped library)
        const u32 prof_on = a.debug != nullptr ? 1u : 0u;
#else
        const u32 prof_on = 0u;
#endif
        const bool tiny = i1 - i0 <= (u64)a.tiny_bytes || i1 < i0;
        const unsigned long long t_begin = a.trace != nullptr ? __builtin_amdgcn_s_memrealtime() : 0ull;
        u32 hand = 0u, hand_level = 0u, late_at = 0u; // hand: 1 = listed for `hand_level` (decoded from its start there), 2 = late list entry `late_at`
        u64 hdr_bitpos = 0ull;
        u64 tstream = prof_on ? (u64)__builtin_readcyclecounter() : 0ull;
#ifdef BRX_BRINGUP
        if (lane < 32u) g_prof[lane] = 0ull;
#endif
        PT_BEGIN(pd);
#if BRX_LEVEL == 0
        if (a.prepass != 0u) {
            // ---- plan B's classification pre-pass: framing + the first compressed meta-block's header, nothing else.  A stream
            // whose tables spill and that has produced no output yet gets the level that holds them; everything else -- fits,
            // errors, streams without a compressed meta-block, output before the first header -- stays with the regular kernel.
            u32 c = 0u;
            u32 st0 = seg_frame();
            if (st0 == SEG_NEED_HEADER) {
                st0 = cold_header();
                if (st0 == ST_OK && get64(s, 20) != 0ull && rfl(s.st[10]) == 0u) c = level_for(rfl(s.st[ST_NEED]));
            }
            {
                const u32 *slab = (const u32 *)(uintptr_t)get64(s, 20);
                if (slab != nullptr) scratch_release(a.pool, slab);
            }
            if (lane == 0u) a.cls[sid] = (u8)c;
            if (c == 0u && lane == 0u) { // (the regular kernel's own streams: their mean size splits its queue, BrxKernelArgs::big_bytes)
                const u64 len = i1 >= i0 ? i1 - i0 : 0ull;
                const u32 units = (u32)(len >> 6 > 0xfffffull ? 0xfffffull : len >> 6);
                (void)atomicAdd(a.work_counter + 12, units);
                (void)atomicAdd(a.work_counter + 13, 1u);
                (void)atomicMax(a.work_counter + 16, units);
                (void)atomicMax(a.work_counter + 17, 0xfffffu - units);
            }
            if (c != 0u) {
                const u32 slot = rdl(atomicAdd(a.work_counter + 4 + c, lane == 0u ? 1u : 0u), 0);
                if (lane == 0u) a.defer[(size_t)(c - 1u) * a.defer_cap + slot] = sid;
                if (lane == 0u && a.handed_seq != nullptr) *a.handed_seq = a.launch_seq; // (pinned host word: "this context meets such streams")
            }
            continue;
        }
#endif
        u32 st;
#if BRX_LEVEL > 0
        if (late_slot != 0xffffffffu) {
            // ---- resume with state (the late list): the stream was under way in a narrower kernel when the header that comes
            // next outgrew it.  Its record goes over the fresh state, the ring comes back from HBM, and the loop below starts at
            // that header.
            const u32 *rec = LATE_IN_RECS + (size_t)late_slot * HU_WORDS;
            const u32 rpos = rfl(rec[HU_POS]);
            if (lane == 0u) {
                s.st[3] = rec[HU_BITPOS]; s.st[4] = rec[HU_BITPOS + 1];
                s.st[10] = rpos; s.st[12] = rpos + s.st[11];
                s.st[13] = rec[HU_WINDOW];
                s.st[14] = rec[HU_DIST]; s.st[15] = rec[HU_DIST + 1]; s.st[16] = rec[HU_DIST + 2]; s.st[17] = rec[HU_DIST + 3];
                s.st[29] = rec[HU_WD]; s.st[30] = rec[HU_WD + 1];
                s.st[ST_STARTED] = 1u; s.st[ST_ISLAST] = rec[HU_ISLAST]; s.st[ST_MLEN] = rec[HU_MLEN];
                (void)atomicSub(a.work_counter + 11, rpos); // nothing of this stream is decoded twice
            }
            seg_resume();
            st = SEG_NEED_HEADER;
        } else
#endif
        st = seg_frame();
        PT_ADD(0, pd);
        while (st == SEG_NEED_HEADER) {
            hdr_bitpos = get64(s, 3);
            st = cold_header();
            PT_ADD(1, pd);
            if (st) break;
#if BRX_LEVEL < BRX_LEVELS - 1
            if (a.defer != nullptr && get64(s, 20) != 0ull) {
                // This meta-block's tables spilled into a slab: a stream for the level that holds them.  With no output so far
                // it is listed for that level and decoded from its start (plan A's classification, done by the regular kernel
                // on its way); otherwise -- and whenever the class lists are already being read (plan B) -- it goes to the
                // late list with its state.  A full late list: the stream stays and runs this meta-block from its slab.
                hand_level = level_for(rfl(s.st[ST_NEED]));
                if (hand_level <= BRX_LEVEL) hand_level = BRX_LEVEL + 1u;
                if (BRX_LEVEL == 0 && a.late_only == 0u && rfl(s.st[10]) == 0u) {
                    hand = 1u;
                } else {
                    if (LATE_OUT_RECS != nullptr) {
                        late_at = rdl(atomicAdd(a.work_counter + LATE_OUT_CNT, lane == 0u ? 1u : 0u), 0);
                        if (late_at < LATE_OUT_CAP) hand = 2u;
                    }
                }
                if (hand != 0u) break;
            }
#endif
            if (a.debug_stop == 8u || (tiny && a.debug_stop == 0u)) {
                // the C++ loop alone, whole meta-block per call: a bring-up mode, and the way of streams of a few dozen
                // bytes (a handful of commands, e.g. the RLE-like fills of BASELINE configs 3 / 4): preparing the
                // assembly loop's tables and handing over at every long copy costs more than it saves there
                st = generic_commands(HC_WHOLE);
                PT_ADD(2, pd);
            } else if (a.debug_stop == 7u) { // bring-up: the C++ loop alone, one command per call
                st = generic_commands(HC_START);
                while (st == HC_CONTINUE) st = generic_commands(HC_RESUME_R1);
#ifdef BRX_BRINGUP
            } else if (a.debug_stop == 9u) { // bring-up: as 7, and every dump_interval-th parked state that the assembly
                                             // loop could be entered with goes to a.dump (input of tools/asm_emu.py)
                st = generic_commands(HC_START);
                u32 k = 0;
                while (st == HC_CONTINUE) {
                    if (lane == 0u) { const u32 we = (u32)(get64(s, 5) >> 5); s.mbw[MBW_WSAFE] = we > BRX_END_MARGIN ? we - BRX_END_MARGIN : 0u; } // (what the emulated loop reads)
                    if (a.dump != nullptr && rfl(s.mbw[MBW_ASM]) != 0u && (k % a.dump_interval) == 0u) {
                        u32 slot = rdl(atomicAdd(a.dump, lane == 0u ? 1u : 0u), 0);
                        if (slot < a.dump_max) {
                            u32 *rec = a.dump + 16u + (size_t)slot * BRX_DUMP_WORDS;
                            if (lane == 0u) { rec[0] = sid; rec[1] = k; }
                            const u32 *src = (const u32 *)&s;
                            for (u32 w = lane; w < BRX_LDS_BYTES / 4u; w += 64u) rec[16u + w] = src[w];
                        }
                    }
                    k++;
                    st = generic_commands(HC_RESUME_R1);
                }
#endif
            } else {
                // The assembly loop (brx_hot.S) with the C++ loop as its safety net.  The C++ side reads the first
                // insert&copy symbol (exact end-of-input rules) and decides whether the meta-block qualifies; then
                // the assembly runs until something unusual comes up, C++ takes exactly one command (or finishes
                // the meta-block), and so on.
                st = generic_commands(HC_START);
                PT_ADD(2, pd);
                const bool use_asm = rfl(s.mbw[MBW_ASM]) != 0u;
                if (prof_on && lane == 0u) s.pad[use_asm ? 2 : 0]++;
                if (st == HC_CONTINUE && !use_asm) st = generic_commands(HC_RESUME_R1_WHOLE);
                // ---- speculative end (round 5; checkpointed in round 6).  In a batch the assembly loop is poisoned only BEHIND the end of
                // the input (brx_hot.S, mbw[MBW_WSAFE]): a valid stream never gets there -- its last meta-block ends first, and its last
                // bytes do not pass through the C++ loop at 2 - 4 k cycles a symbol (a third of a 400-byte stream's time).  A stream that
                // DID consume bits beyond its end (truncated, or corrupted so that it reads on) has decoded garbage from some point on:
                // whatever status came of it is void.  Round 5 took the WHOLE meta-block back and ran it through the C++ loop -- a cut
                // 1 MiB single-meta-block stream cost 0.35 s next to neighbours that take 45 ms.  Now the loop of a
                // long stream is poisoned TWICE: first BRX_SPEC_CK_DWORDS in front of the end -- where everything consumed was real --
                // and the dispatcher keeps the parked state of that moment (Lds::st and Lds::mbw, one word per lane of two registers:
                // the checkpoint), then behind the end as before.  A stream that ran on goes back to the checkpoint -- everything out
                // to HBM, st / mbw as they stood, the ring reloaded from the stream's own output as for a late resume -- and finishes
                // with the margin IN FRONT of the end and the C++ loop's exact rules for the tail: what is decoded twice is those few
                // dozen dwords.  Streams of a few KiB keep one stage (the checkpoint is the first entry of the loop): an extra
                // hand-over per stream would cost them what the speculation saves.
                const u32 w_end = (u32)(get64(s, 5) >> 5);
                bool spec = use_asm && (rfl(s.st[ST_SPEC]) & 1u) != 0u;
                bool have_ck = false;
                u32 ck_st = 0u, ck_mbw = 0u;
                for (;;) {
                    while (st == HC_CONTINUE) {
                        PT_ADD(3, pd);
                        u32 wsafe = w_end > BRX_END_MARGIN ? w_end - BRX_END_MARGIN : 0u; // (no speculation: every bit consumed in the loop is a real one)
                        bool stage1 = false;
                        if (spec) {
                            const u32 cw = (u32)(get64(s, 3) >> 5);
                            if (!have_ck && cw + BRX_SPEC_CK_MIN_DWORDS < w_end) {
                                stage1 = true;
                                wsafe = w_end - BRX_SPEC_CK_DWORDS;
                            } else {
                                if (!have_ck) {
                                    ck_st = s.st[lane < 48u ? lane : 0u];
                                    ck_mbw = s.mbw[lane < 48u ? lane : 0u];
                                    have_ck = true;
                                }
                                wsafe = w_end + 4u;
                            }
                        }
                        if (lane == 0u) s.mbw[MBW_WSAFE] = wsafe;
                        const u32 r = sw_loop ? asm_commands_sw() : asm_commands();
                        PT_ADD(4, pd);
                        if (prof_on && lane == 0u) {
                            s.pad[4 + (r & 3u)]++;
                        }
                        // (the loop came back with its cursor BEYOND the end: everything since the checkpoint is void -- and the C++ loop must
                        // not see this state: "bits left" = end - cursor wraps, and it would decode zeros to the end of the meta-block at
                        // ~3 000 cycles a symbol: that, not the redo, was most of a cut stream's time)
                        if (spec && have_ck && get64(s, 3) > get64(s, 5)) break;
                        // (the loop left because the meta-block is complete -- R1 with nothing left: the C++ side would load the whole state
                        // only to find that out; ~8 k cycles per meta-block, 2 - 4 % of a stream flushed every KiB)
#ifndef BRX_NO_SKIP_END
                        if ((r & 3u) == 1u && rfl(s.mbw[MBW_MBLEFT]) == 0u) { st = ST_OK; break; }
#endif
                        // (bit 4 of the exit word: "the cursor is at the poison point, the loop would only hand straight back" -- true of the
                        // last one; behind the first one the loop goes on, with the checkpoint taken)
                        st = generic_commands((HC_RESUME_R0 + ((r & 3u) > 2u ? 1u : (r & 3u))) | (((r & 16u) && !stage1) ? HC_TO_END : 0u));
                    }
                    PT_ADD(3, pd);
                    if (!(spec && have_ck && get64(s, 3) > get64(s, 5))) break;
                    if (lane == 0u) (void)atomicAdd(a.work_counter + 18, 1u);
                    seg_finish();
                    __threadfence();
                    if (lane < 48u && lane != 20u && lane != 21u) s.st[lane] = ck_st; // (a slab claimed meanwhile stays claimed)
                    if (lane < 48u) s.mbw[lane] = ck_mbw;
#ifdef BRX_CK_VFL_POS
                    if (lane == 0u) s.st[12] = s.st[10] + s.st[11];
#endif
                    seg_resume();
                    spec = false;
                    st = HC_CONTINUE;
                }
            }
            if (st) break;
            st = seg_frame();
            PT_ADD(0, pd);
        }
        seg_finish();
        PT_ADD(5, pd);
        {
            const u32 *slab = (const u32 *)(uintptr_t)get64(s, 20);
            if (slab != nullptr) scratch_release(a.pool, slab);
        }
#if BRX_LEVEL < BRX_LEVELS - 1
        if (hand != 0u) { // no status, no length: a wider level finishes the stream (same bytes, same slots)
            if (hand == 1u) {
                const u32 slot = rdl(atomicAdd(a.work_counter + 4 + hand_level, lane == 0u ? 1u : 0u), 0);
                if (lane == 0u) a.defer[(size_t)(hand_level - 1u) * a.defer_cap + slot] = sid;
            } else if (lane == 0u) { // (everything decoded so far is in HBM: seg_finish above)
                u32 *rec = LATE_OUT_RECS + (size_t)late_at * HU_WORDS;
                rec[HU_BITPOS] = (u32)hdr_bitpos; rec[HU_BITPOS + 1] = (u32)(hdr_bitpos >> 32);
                rec[HU_POS] = s.st[10]; rec[HU_WINDOW] = s.st[13];
                rec[HU_DIST] = s.st[14]; rec[HU_DIST + 1] = s.st[15]; rec[HU_DIST + 2] = s.st[16]; rec[HU_DIST + 3] = s.st[17];
                rec[HU_WD] = s.st[29]; rec[HU_WD + 1] = s.st[30];
                rec[HU_ISLAST] = s.st[ST_ISLAST]; rec[HU_MLEN] = s.st[ST_MLEN]; rec[HU_SID] = sid;
#if BRX_LEVEL < 3
                a.defer[3u * (size_t)a.defer_cap + late_at] = sid;
#endif
                (void)atomicAdd(a.work_counter + 11, s.st[10]); // taken back by the resume: what stays was decoded twice
            }
            if (lane == 0u && a.handed_seq != nullptr) *a.handed_seq = a.launch_seq; // (pinned host word: "this context meets such streams")
            continue;
        }
#endif
        u32 pos = rfl(s.st[10]), needed = rfl(s.st[22]);
        if (prof_on && lane < 8u) a.debug[(size_t)sid * 10u + lane] = (u64)s.pad[2 * lane] | ((u64)s.pad[2 * lane + 1] << 32);
        if (prof_on && lane == 0u) {
            a.debug[(size_t)sid * 10u + 8] = (u64)__builtin_readcyclecounter() - tstream;
            a.debug[(size_t)sid * 10u + 9] = s.st[19];
        }
#ifdef BRX_BRINGUP
        if (prof_on && lane < 32u) a.debug[(size_t)a.n_total * 10u + (size_t)sid * 32u + lane] = g_prof[lane];
#endif
        if (lane == 0u) {
            a.status[sid] = (int)final_status(st);
            a.out_len[sid] = st == ST_OUTPUT_TOO_SMALL ? (u64)needed : (u64)pos;
        }
        if (a.trace != nullptr && lane == 0u) {
            a.trace[(size_t)sid * 4u] = t_begin;
            a.trace[(size_t)sid * 4u + 1u] = __builtin_amdgcn_s_memrealtime();
            a.trace[(size_t)sid * 4u + 2u] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)BRX_LEVEL << 32);
            a.trace[(size_t)sid * 4u + 3u] = (unsigned long long)blockIdx.x | ((unsigned long long)gridDim.x << 32);
        }
    }
}

#undef generic_commands

void BRX_LAUNCH_NAME(const BrxKernelArgs &args, unsigned grid, void *hip_stream) {
    hipLaunchKernelGGL(BRX_KERNEL_NAME, dim3(grid), dim3(BRX_WAVE), 0, (hipStream_t)hip_stream, args);
}
#else // BRX_SMALL: the lean instance
#include "brx_small.h"
#endif
