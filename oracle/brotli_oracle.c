/*
 * brotli_oracle.c -- CPU restatement of the decode path of ende76/brotli-rs
 * (crate `brotli` v0.3.23).  TEST INFRASTRUCTURE ONLY -- see brotli_oracle.h.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference tree).  Structure is deliberately different from the reference
 * (straight-line parser over an in-memory buffer, the output buffer doubles as
 * the sliding window) but every observable -- output bytes for valid streams,
 * error kind for invalid ones -- follows the reference, quirks Q1..Q15 of
 * SURVEY.md section 2.3 included.
 *
 * Constant tables come from the format specification text (tools/gen_tables.py,
 * CRC-checked), not from the reference's .rs files.
 */
#include "brotli_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int g_trace;

#include "_gen/tables_gen.h" /* BRO_DICT, BRO_CONTEXT_LUT, BRO_TRANSFORMS (tools/bin2h.py) */

/* ------------------------------------------------------------------------- */
/* Bit reader: restates src/bitreader/mod.rs:21-303.                          */
/* The reference keeps (current_byte, bit_pos) over a BufReader; every method */
/* there is equivalent to "take n bits LSB-first at a global bit position,    */
/* fail if fewer than n remain" (read_bit :178-203, read_u8 :58-84,           */
/* read_u8_from_nibble :88-135, read_uN_from_n_bits :140-158/:237-253/        */
/* :272-288, read_u8_from_byte_tail :257-267, read_fixed_length_string        */
/* :292-303).  The state (bit_pos==0, current_byte==Some) never arises, so the */
/* odd arms at :67-71/:77-81 are unreachable.                                 */
/* ------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *p;
    uint64_t nbits; /* 8 * input length */
    uint64_t pos;   /* bits consumed */
} BR;

static inline int br_bit(BR *r) { /* -1 = end of input */
    if (r->pos >= r->nbits) return -1;
    int b = (r->p[r->pos >> 3] >> (r->pos & 7)) & 1;
    r->pos++;
    return b;
}

/* n <= 32.  Returns 0, or -1 when fewer than n bits remain (bits up to the end are consumed,
 * like the bit-at-a-time loops of the reference; irrelevant since every caller aborts). */
static inline int br_bits(BR *r, unsigned n, uint32_t *v) {
    if (r->pos + n > r->nbits) {
        r->pos = r->nbits;
        return -1;
    }
    uint64_t acc = 0;
    uint64_t byte = r->pos >> 3;
    unsigned sh = (unsigned)(r->pos & 7);
    unsigned need = (sh + n + 7) >> 3;
    for (unsigned i = 0; i < need; i++) acc |= (uint64_t)r->p[byte + i] << (8 * i);
    *v = (uint32_t)((acc >> sh) & ((n == 32) ? 0xffffffffull : ((1ull << n) - 1)));
    r->pos += n;
    return 0;
}

/* Up to 16 bits at the cursor without consuming, zero padded past the end. */
static inline uint32_t br_peek16(const BR *r) {
    uint64_t byte = r->pos >> 3;
    uint64_t nbytes = r->nbits >> 3;
    uint32_t acc = 0;
    for (unsigned i = 0; i < 3; i++)
        if (byte + i < nbytes) acc |= (uint32_t)r->p[byte + i] << (8 * i);
    return (acc >> (r->pos & 7)) & 0xffff;
}

/* read_u8_from_byte_tail, src/bitreader/mod.rs:257-267: the bits up to the next byte boundary. */
static inline uint32_t br_byte_tail(BR *r) {
    unsigned k = (unsigned)(r->pos & 7);
    uint32_t v = 0;
    if (k) br_bits(r, 8 - k, &v); /* cannot fail: those bits belong to a byte that exists */
    return v;
}

/* ------------------------------------------------------------------------- */
/* Prefix codes: restates src/huffman/mod.rs:19-49 (canonical assignment, Q7) */
/* and src/huffman/tree/mod.rs:34-93 (heap-array tree, lookup, Q5, Q15).      */
/* ------------------------------------------------------------------------- */
#define BRO_MAX_ALPHABET 704

typedef struct {
    uint16_t nsym;        /* Tree.len: number of inserted symbols */
    uint16_t last_symbol; /* Tree.last_symbol */
    uint8_t max_len;      /* max code length = tree depth */
    uint16_t count[16];   /* codes per length */
    uint16_t first[16];   /* first code of each length, already masked to `len` bits (Q7) */
    uint16_t offs[16];    /* start of that length's run in sorted[] */
    uint16_t *sorted;     /* symbols in (length, insertion order) */
    uint16_t root[256];   /* 8-bit first-level table: (symbol << 4) | len, 0 = not resolved in 8 bits */
    int32_t *heap;        /* BRO_FLAG_TREE_WALK: Vec<Option<u16>> of size 2^(max_len+1)-1, -1 = None */
    uint32_t heap_len;
} PCode;

static void pcode_free(PCode *c) {
    free(c->sorted);
    free(c->heap);
    c->sorted = NULL;
    c->heap = NULL;
}

static unsigned rev_bits(unsigned v, unsigned n) {
    unsigned r = 0;
    for (unsigned i = 0; i < n; i++) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}

/* codes_from_lengths_and_symbols, src/huffman/mod.rs:19-43.  bl_count[0] counts the zero-length
 * symbols (Q7); bit_string_from_code_and_length (:3-11) keeps only the low `len` bits of the code,
 * which is what makes that harmless -- reproduced by the mask below. */
static int pcode_build(PCode *c, const uint8_t *lengths, const uint16_t *symbols, unsigned n, unsigned flags) {
    memset(c, 0, sizeof *c);
    unsigned max_len = 0;
    unsigned bl_count[16] = {0};
    for (unsigned i = 0; i < n; i++) {
        if (lengths[i] > max_len) max_len = lengths[i];
        bl_count[lengths[i]]++;
    }
    unsigned next_code[16] = {0};
    unsigned code = 0;
    for (unsigned bits = 1; bits <= max_len; bits++) {
        code = (code + bl_count[bits - 1]) << 1;
        next_code[bits] = code;
    }
    c->max_len = (uint8_t)max_len;
    c->sorted = (uint16_t *)malloc(sizeof(uint16_t) * (n ? n : 1));
    if (!c->sorted) return -1;
    unsigned off = 0;
    for (unsigned l = 0; l <= max_len; l++) {
        c->offs[l] = (uint16_t)off;
        c->first[l] = (uint16_t)(next_code[l] & ((1u << l) - 1));
        if (l > 0 || max_len == 0) off += bl_count[l];
    }
    unsigned fill[16] = {0};
    if (flags & BRO_FLAG_TREE_WALK) { /* Tree::with_max_depth, src/huffman/tree/mod.rs:34-40 */
        c->heap_len = (1u << (max_len + 1)) - 1;
        c->heap = (int32_t *)malloc(sizeof(int32_t) * c->heap_len);
        if (!c->heap) return -1;
        for (uint32_t i = 0; i < c->heap_len; i++) c->heap[i] = -1;
    }
    for (unsigned i = 0; i < n; i++) {
        unsigned len = lengths[i];
        if (len > 0 || max_len == 0) { /* src/huffman/mod.rs:36 */
            unsigned cd = (next_code[len] + fill[len]) & ((1u << len) - 1);
            c->sorted[c->offs[len] + fill[len]] = symbols[i];
            fill[len]++;
            c->count[len]++;
            c->nsym++;
            c->last_symbol = symbols[i];
            if (c->heap) c->heap[((1u << len) - 1) + cd] = symbols[i]; /* Tree::insert :50-61 */
            if (len >= 1 && len <= 8) {
                unsigned r = rev_bits(cd, len);
                for (unsigned w = r; w < 256; w += 1u << len) c->root[w] = (uint16_t)((symbols[i] << 4) | len);
            }
        }
    }
    if (g_trace) { /* analysis aid: alphabet size, symbols present, codes per length */
        fprintf(stderr, "TREE %u %u %u :", n, (unsigned)c->nsym, max_len);
        for (unsigned l = 1; l <= max_len; l++) fprintf(stderr, " %u", (unsigned)c->count[l]);
        fprintf(stderr, "\n");
    }
    return 0;
}

enum { LK_OK = 0, LK_NONE = 1, LK_EOF = 2 };

/* Tree::lookup_symbol / Tree::lookup, src/huffman/tree/mod.rs:63-93.
 *   len==0 -> Ok(None); len==1 -> last_symbol with ZERO bits consumed (Q5);
 *   otherwise walk one bit per level; walking off the array (max_len+1 bits read) -> Ok(None) (Q15);
 *   running out of input first -> Err. */
static inline int pcode_lookup(const PCode *c, BR *r, unsigned *sym) {
    if (c->nsym == 0) return LK_NONE;
    if (c->nsym == 1) {
        *sym = c->last_symbol;
        return LK_OK;
    }
    if (c->heap) { /* structure-faithful walk */
        uint32_t pseudo = 1;
        for (;;) {
            int b = br_bit(r);
            if (b < 0) return LK_EOF;
            pseudo = (pseudo << 1) + (uint32_t)b;
            uint32_t idx = pseudo - 1;
            if (idx > c->heap_len - 1) return LK_NONE;
            if (c->heap[idx] >= 0) {
                *sym = (unsigned)c->heap[idx];
                return LK_OK;
            }
        }
    }
    /* Equivalent canonical decoder.  Fast path: 8-bit root table on zero-padded peeked bits; an entry
     * is only trusted when all of its bits really exist. */
    uint64_t remain = r->nbits - r->pos;
    uint32_t peek = br_peek16(r);
    unsigned e = c->root[peek & 0xff];
    if (e && (e & 15u) <= remain) {
        *sym = e >> 4;
        r->pos += e & 15u;
        return LK_OK;
    }
    unsigned code = 0;
    for (unsigned len = 1; len <= c->max_len; len++) {
        int b = br_bit(r);
        if (b < 0) return LK_EOF;
        code = (code << 1) | (unsigned)b;
        unsigned d = code - c->first[len];
        if (code >= c->first[len] && d < c->count[len]) {
            *sym = c->sorted[c->offs[len] + d];
            return LK_OK;
        }
    }
    if (br_bit(r) < 0) return LK_EOF; /* the reference reads one more bit before falling off the array */
    return LK_NONE;
}

/* ------------------------------------------------------------------------- */
/* Closed-form tables.                                                        */
/* ------------------------------------------------------------------------- */
/* NDBITS / DOFFSET: spec section 8 (docs/draft-alakuijala-brotli-07.txt:1411-1425);
 * same values as src/dictionary/mod.rs:1-11. */
static const uint8_t NDBITS[25] = {0, 0, 0, 0, 10, 10, 11, 11, 10, 10, 10, 10, 10, 9, 9, 8, 7, 7, 8, 7, 7, 6, 6, 5, 5};
static uint32_t DOFFSET[25];

/* Insert / copy length codes: spec section 5 (docs/draft...:961-1070); generator recipe kept as a
 * comment in src/lookuptable/mod.rs:59-121; the constant is src/lookuptable/mod.rs:123. */
static const uint16_t INS_BASE[24] = {0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26, 34, 50, 66, 98, 130, 194, 322, 578, 1090, 2114, 6210, 22594};
static const uint8_t INS_EXTRA[24] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24};
static const uint16_t COPY_BASE[24] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 18, 22, 30, 38, 54, 70, 102, 134, 198, 326, 582, 1094, 2118};
static const uint8_t COPY_EXTRA[24] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24};
static const uint8_t CELL_INS[11] = {0, 0, 0, 0, 8, 8, 0, 16, 8, 16, 16};
static const uint8_t CELL_COPY[11] = {0, 8, 0, 8, 0, 8, 16, 0, 16, 8, 16};

void bro_insert_copy_entry(unsigned sym, uint32_t *ib, uint32_t *ie, uint32_t *cb, uint32_t *ce) {
    unsigned cell = sym >> 6;
    unsigned icode = CELL_INS[cell] + ((sym >> 3) & 7);
    unsigned ccode = CELL_COPY[cell] + (sym & 7);
    *ib = INS_BASE[icode];
    *ie = INS_EXTRA[icode];
    *cb = COPY_BASE[ccode];
    *ce = COPY_EXTRA[ccode];
}

/* Block count code, src/lib.rs:962-976. */
static const uint16_t BLEN_BASE[26] = {1, 5, 9, 13, 17, 25, 33, 41, 49, 65, 81, 97, 113, 145, 177, 209, 241, 305, 369, 497, 753, 1265, 2289, 4337, 8433, 16625};
static const uint8_t BLEN_EXTRA[26] = {2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 7, 8, 9, 10, 11, 12, 13, 24};

/* 121 transforms parsed out of the CRC-checked spec blob (prefix\0 op suffix\0). */
typedef struct {
    const uint8_t *prefix;
    const uint8_t *suffix;
    uint8_t plen, slen, op;
} Xform;
static Xform XFORMS[121];
static int g_init_done;

static void bro_init(void) {
    if (g_init_done) return;
    DOFFSET[0] = 0;
    for (unsigned l = 0; l < 24; l++) DOFFSET[l + 1] = DOFFSET[l] + (l >= 4 ? l * (1u << NDBITS[l]) : 0);
    const uint8_t *p = BRO_TRANSFORMS;
    for (unsigned i = 0; i < 121; i++) {
        XFORMS[i].prefix = p;
        XFORMS[i].plen = (uint8_t)strlen((const char *)p);
        p += XFORMS[i].plen + 1;
        XFORMS[i].op = *p++;
        XFORMS[i].suffix = p;
        XFORMS[i].slen = (uint8_t)strlen((const char *)p);
        p += XFORMS[i].slen + 1;
    }
    g_init_done = 1;
}

const uint8_t *bro_dictionary(void) { return BRO_DICT; }
const uint8_t *bro_context_lut(int which) { return BRO_CONTEXT_LUT + 256 * which; }

/* ------------------------------------------------------------------------- */
/* Word transformations: restates src/transformation/mod.rs:3-209.            */
/* ------------------------------------------------------------------------- */
/* uppercase_all, :3-40 */
static unsigned xf_uppercase_all(const uint8_t *w, unsigned l, uint8_t *o) {
    unsigned i = 0, n = 0;
    while (i < l) {
        uint8_t b = w[i];
        if (b <= 96 || (b >= 123 && b <= 191)) {
            o[n++] = b;
            i += 1;
        } else if (b <= 122) {
            o[n++] = b ^ 32;
            i += 1;
        } else if (b <= 223) {
            o[n++] = b;
            if (i + 1 < l) o[n++] = w[i + 1] ^ 32;
            i += 2;
        } else {
            o[n++] = b;
            if (i + 1 < l) o[n++] = w[i + 1];
            if (i + 2 < l) o[n++] = w[i + 2] ^ 5;
            i += 3;
        }
    }
    return n;
}

/* uppercase_first, :42-82.  Returns -1 for a first byte of 0x00: the reference has no match arm
 * for it (`1...96|123...191`) and hits unreachable!() -> panic (Q3). */
static int xf_uppercase_first(const uint8_t *w, unsigned l, uint8_t *o) {
    if (l == 0) return 0;
    unsigned n = 0, i;
    uint8_t b = w[0];
    if (b == 0) return -1;
    if (b <= 96 || (b >= 123 && b <= 191)) {
        o[n++] = b;
        i = 1;
    } else if (b <= 122) {
        o[n++] = b ^ 32;
        i = 1;
    } else if (b <= 223) {
        o[n++] = b;
        if (1 < l) o[n++] = w[1] ^ 32;
        i = 2;
    } else {
        o[n++] = b;
        if (1 < l) o[n++] = w[1];
        if (2 < l) o[n++] = w[2] ^ 5;
        i = 3;
    }
    /* `&base_word[i..]` with i > len panics in Rust too; unreachable for dictionary words (len >= 4). */
    for (; i < l; i++) o[n++] = w[i];
    return (int)n;
}

/* transformation, :84-209: prefix + elementary op + suffix. */
int bro_transform(unsigned id, const uint8_t *word, unsigned len, uint8_t *out) {
    bro_init();
    if (id > 120) return -1;
    const Xform *x = &XFORMS[id];
    unsigned n = 0;
    memcpy(out, x->prefix, x->plen);
    n += x->plen;
    unsigned op = x->op;
    if (op == 0) {
        memcpy(out + n, word, len);
        n += len;
    } else if (op == 1) {
        int k = xf_uppercase_first(word, len, out + n);
        if (k < 0) return -1;
        n += (unsigned)k;
    } else if (op == 2) {
        n += xf_uppercase_all(word, len, out + n);
    } else if (op <= 11) { /* OmitFirstN: &base_word[min(N, len-1)..]  (Q1: keeps the last byte when N >= len) */
        unsigned N = op - 2;
        unsigned from = (len == 0) ? 0 : (N < len - 1 ? N : len - 1);
        memcpy(out + n, word + from, len - from);
        n += len - from;
    } else { /* OmitLastN: &base_word[..max(N, len) - N] */
        unsigned N = op - 11;
        unsigned keep = (len > N ? len : N) - N;
        memcpy(out + n, word, keep);
        n += keep;
    }
    memcpy(out + n, x->suffix, x->slen);
    n += x->slen;
    return (int)n;
}

/* inverse_move_to_front_transform, src/lib.rs:1164-1177. */
void bro_inverse_mtf(uint8_t *v, size_t n) {
    uint8_t mtf[256];
    for (unsigned i = 0; i < 256; i++) mtf[i] = (uint8_t)i;
    for (size_t k = 0; k < n; k++) {
        unsigned index = v[k];
        uint8_t value = mtf[index];
        v[k] = value;
        for (unsigned j = index; j >= 1; j--) mtf[j] = mtf[j - 1];
        mtf[0] = value;
    }
}

/* ------------------------------------------------------------------------- */
/* Decoder state                                                              */
/* ------------------------------------------------------------------------- */
#define BLEN_INF 0xffffffffu /* Option::None block length: NBLTYPES == 1 (Q12) */

typedef struct {
    unsigned nbl;   /* NBLTYPES */
    unsigned btype, btype_prev;
    uint32_t blen;  /* remaining count, BLEN_INF = None */
    PCode types, counts;
    int have_codes;
} BlockCat;

typedef struct {
    BR br;
    uint8_t *out;
    size_t cap;
    size_t pos;            /* Decompressor.count_output, src/lib.rs:385 */
    size_t window;         /* Header.window_size = (1<<WBITS)-16, src/lib.rs:1562 */
    uint8_t p1, p2;        /* literal_buf: last two bytes, src/lib.rs:389,407 */
    uint32_t dist[4];      /* distance_buf, dist[0] = last, src/lib.rs:393,408 */
    unsigned flags;
    size_t needed;         /* for BRO_OUTPUT_TOO_SMALL */
    bro_stats st;
} Dec;

static const char *const STATUS_STR[28] = {
    "OK",
    "Code length check sum did not add up in complex prefix code",
    "Expected end-of-stream, but stream did not end",
    "More uncompressed bytes than expected in meta-block",
    "Encountered invalid value for block count code",
    "Encountered invalid value for block switch command code",
    "Encountered invalid length in reference to static dictionary",
    "Most significant byte of MSKIPLEN was zero",
    "Encountered invalid symbol in prefix code",
    "Encountered invalid transform id in reference to static dictionary",
    "Encountered invalid non-positive distance",
    "Encountered invalid complex prefix code with less than two non-zero codelengths",
    "Encountered invalid complex prefix code with all zero codelengths",
    "Enocuntered non-zero fill bit",
    "Enocuntered non-zero reserved bit",
    "Enocuntered non-zero bit trailing the stream",
    "Enocuntered non-zero nibble trailing",
    "Error parsing context map",
    "Error parsing code lengths for complex prefix code",
    "Error parsing DistanceCode",
    "Error parsing Insert And Copy Length",
    "Error parsing Insert Literals",
    "Error accessing distance ring buffer",
    "Run length excceeded declared length of context map",
    "Encountered unexpected EOF",
    "Output capacity too small",
    "Reference implementation would panic (UppercaseFirst on a word starting with 0x00)",
    "Internal decode-loop watchdog tripped",
};

const char *bro_status_str(int s) { return (s >= 0 && s <= 27) ? STATUS_STR[s] : "unknown status"; }

/* ------------------------------------------------------------------------- */
/* Header pieces                                                              */
/* ------------------------------------------------------------------------- */
/* parse_wbits, src/lib.rs:412-418 over the fixed tree src/lib.rs:89-119.  Code (stream order):
 * 0 -> 16; 1 nnn (n!=0, LSB first) -> 17+n; 1 000 mmm: m=0 -> 17, m=1 -> no entry (the walk reads an
 * 8th bit and falls off the 255-entry array: Ok(None)), m>=2 -> 8+m.  None and Err both map to
 * UnexpectedEOF. */
static int parse_wbits(Dec *d, unsigned *wbits) {
    uint32_t v;
    int b = br_bit(&d->br);
    if (b < 0) return BRO_UNEXPECTED_EOF;
    if (b == 0) {
        *wbits = 16;
        return 0;
    }
    if (br_bits(&d->br, 3, &v)) return BRO_UNEXPECTED_EOF;
    if (v != 0) {
        *wbits = 17 + v;
        return 0;
    }
    if (br_bits(&d->br, 3, &v)) return BRO_UNEXPECTED_EOF;
    if (v == 1) return BRO_UNEXPECTED_EOF;
    *wbits = (v == 0) ? 17 : 8 + v;
    return 0;
}

/* parse_n_bltypes, src/lib.rs:501-525 over the fixed tree src/lib.rs:126-132.  Also used for NTREESL /
 * NTREESD (:575-587).  Code: 0 -> 1; 1 kkk: k=0 -> 2, else (1<<k)+1 plus k extra bits. */
static int parse_n_bltypes(Dec *d, unsigned *n) {
    uint32_t k, extra;
    int b = br_bit(&d->br);
    if (b < 0) return BRO_UNEXPECTED_EOF;
    if (b == 0) {
        *n = 1;
        return 0;
    }
    if (br_bits(&d->br, 3, &k)) return BRO_UNEXPECTED_EOF;
    if (k == 0) {
        *n = 2;
        return 0;
    }
    if (br_bits(&d->br, k, &extra)) return BRO_UNEXPECTED_EOF;
    *n = (1u << k) + 1 + extra;
    return 0;
}

/* parse_simple_prefix_code, src/lib.rs:597-665 (Q8). */
static int parse_simple_prefix_code(Dec *d, unsigned alphabet, PCode *pc) {
    unsigned bit_width = 0;
    while ((1u << bit_width) < alphabet) bit_width++; /* 16 - leading_zeros(alphabet-1), :598 */
    uint32_t v;
    if (br_bits(&d->br, 2, &v)) return BRO_UNEXPECTED_EOF;
    unsigned nsym = v + 1;
    uint16_t s[4];
    for (unsigned i = 0; i < nsym; i++) {
        if (br_bits(&d->br, bit_width, &v)) return BRO_UNEXPECTED_EOF;
        if (v >= alphabet) return BRO_INVALID_SYMBOL;
        s[i] = (uint16_t)v;
    }
    for (unsigned i = 0; i + 1 < nsym; i++)
        for (unsigned j = i + 1; j < nsym; j++)
            if (s[i] == s[j]) return BRO_INVALID_SYMBOL;
    int tree_select = 0;
    if (nsym == 4) {
        tree_select = br_bit(&d->br);
        if (tree_select < 0) return BRO_UNEXPECTED_EOF;
    }
    uint8_t len[4];
#define SORT2(a, b) do { if (s[a] > s[b]) { uint16_t t = s[a]; s[a] = s[b]; s[b] = t; } } while (0)
    switch (nsym) {
    case 1: len[0] = 0; break;
    case 2: SORT2(0, 1); len[0] = 1; len[1] = 1; break;
    case 3: SORT2(1, 2); len[0] = 1; len[1] = 2; len[2] = 2; break;
    default:
        if (!tree_select) {
            SORT2(0, 1); SORT2(2, 3); SORT2(0, 2); SORT2(1, 3); SORT2(1, 2);
            len[0] = len[1] = len[2] = len[3] = 2;
        } else {
            SORT2(2, 3);
            len[0] = 1; len[1] = 2; len[2] = 3; len[3] = 3;
        }
    }
#undef SORT2
    return pcode_build(pc, len, s, nsym, d->flags) ? -1 : 0;
}

/* The fixed code-length code, src/lib.rs:120-125 (stream order): 00->0 01->3 10->4 110->2 1110->1 1111->5. */
static int read_code_length_code(Dec *d, unsigned *v) {
    int b0 = br_bit(&d->br);
    if (b0 < 0) return -1;
    int b1 = br_bit(&d->br);
    if (b1 < 0) return -1;
    if (!b0) {
        *v = b1 ? 3 : 0;
        return 0;
    }
    if (!b1) {
        *v = 4;
        return 0;
    }
    int b2 = br_bit(&d->br);
    if (b2 < 0) return -1;
    if (!b2) {
        *v = 2;
        return 0;
    }
    int b3 = br_bit(&d->br);
    if (b3 < 0) return -1;
    *v = b3 ? 5 : 1;
    return 0;
}

/* parse_complex_prefix_code, src/lib.rs:667-875 (Q6, Q15). */
static int parse_complex_prefix_code(Dec *d, unsigned hskip, unsigned alphabet, PCode *pc) {
    static const uint8_t ORDER[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
    uint8_t cl[18] = {0}; /* indexed by code-length symbol, i.e. already permuted like :719 */
    unsigned sum = 0, nonzero = 0;
    for (unsigned i = hskip; i < 18; i++) {
        unsigned v;
        if (read_code_length_code(d, &v)) return BRO_UNEXPECTED_EOF;
        cl[ORDER[i]] = (uint8_t)v;
        if (v > 0) {
            sum += 32u >> v;
            nonzero++;
            if (sum == 32) break;
            if (sum > 32) return BRO_CODE_LENGTHS_CHECKSUM;
        }
    }
    if (nonzero == 0) return BRO_NO_CODE_LENGTH;
    if (nonzero >= 2 && sum < 32) return BRO_CODE_LENGTHS_CHECKSUM;

    PCode clc;
    uint16_t syms18[18];
    for (unsigned i = 0; i < 18; i++) syms18[i] = (uint16_t)i;
    if (pcode_build(&clc, cl, syms18, 18, d->flags)) return -1;

    uint8_t lens[BRO_MAX_ALPHABET];
    memset(lens, 0, alphabet);
    unsigned total = 0;
    int last_symbol = -1;   /* Option<u16> last_symbol */
    unsigned last_repeat = 0; /* Option<usize>, 0 = None (a real repeat is >= 3) */
    unsigned last_nz = 8;
    unsigned i = 0;
    int rc = 0;
    while (i < alphabet) {
        unsigned sym;
        int lk = pcode_lookup(&clc, &d->br, &sym);
        if (lk == LK_EOF) { rc = BRO_UNEXPECTED_EOF; goto done; }
        if (lk == LK_NONE) { rc = BRO_PARSE_ERROR_COMPLEX_PREFIX_CODE_LENGTHS; goto done; }
        if (sym <= 15) {
            lens[i++] = (uint8_t)sym;
            last_symbol = (int)sym;
            last_repeat = 0;
            if (sym > 0) {
                last_nz = sym;
                total += 32768u >> sym;
                if (total == 32768) break;
                if (total > 32768) { rc = BRO_CODE_LENGTHS_CHECKSUM; goto done; }
            }
        } else if (sym == 16) {
            uint32_t extra;
            if (br_bits(&d->br, 2, &extra)) { rc = BRO_UNEXPECTED_EOF; goto done; }
            unsigned add, new_repeat;
            if (last_symbol == 16 && last_repeat) {
                new_repeat = 4 * (last_repeat - 2) + extra + 3;
                add = new_repeat - last_repeat;
            } else {
                new_repeat = 3 + extra;
                add = new_repeat;
            }
            if (i + add > alphabet) { rc = BRO_PARSE_ERROR_COMPLEX_PREFIX_CODE_LENGTHS; goto done; }
            for (unsigned k = 0; k < add; k++) {
                lens[i++] = (uint8_t)last_nz;
                total += 32768u >> last_nz;
            }
            if (total == 32768) break;
            if (total > 32768) { rc = BRO_CODE_LENGTHS_CHECKSUM; goto done; }
            last_repeat = new_repeat;
            last_symbol = 16;
        } else { /* 17 */
            uint32_t extra;
            if (br_bits(&d->br, 3, &extra)) { rc = BRO_UNEXPECTED_EOF; goto done; }
            if (last_symbol == 17 && last_repeat) {
                unsigned new_repeat = 8 * (last_repeat - 2) + extra + 3;
                i += new_repeat - last_repeat;
                last_repeat = new_repeat;
            } else {
                last_repeat = 3 + extra;
                i += last_repeat;
            }
            if (i > alphabet) { rc = BRO_PARSE_ERROR_COMPLEX_PREFIX_CODE_LENGTHS; goto done; }
            last_symbol = 17;
        }
    }
    {
        unsigned nz = 0;
        for (unsigned k = 0; k < alphabet; k++) nz += lens[k] > 0;
        if (nz < 2) { rc = BRO_LESS_THAN_TWO_NON_ZERO_CODE_LENGTHS; goto done; }
        uint16_t *syms = (uint16_t *)malloc(sizeof(uint16_t) * alphabet);
        if (!syms) { rc = -1; goto done; }
        for (unsigned k = 0; k < alphabet; k++) syms[k] = (uint16_t)k;
        rc = pcode_build(pc, lens, syms, alphabet, d->flags) ? -1 : 0; /* codes_from_lengths, src/huffman/mod.rs:45-49 */
        free(syms);
    }
done:
    pcode_free(&clc);
    return rc;
}

/* parse_prefix_code (+ parse_prefix_code_kind), src/lib.rs:589-595, 877-889. */
static int parse_prefix_code(Dec *d, unsigned alphabet, PCode *pc) {
    uint32_t kind;
    memset(pc, 0, sizeof *pc);
    if (br_bits(&d->br, 2, &kind)) return BRO_UNEXPECTED_EOF;
    if (kind == 1) return parse_simple_prefix_code(d, alphabet, pc);
    return parse_complex_prefix_code(d, kind, alphabet, pc);
}

/* parse_block_count, src/lib.rs:957-987.  Ok(None) maps to UnexpectedEOF here (:977). */
static int parse_block_count(Dec *d, const PCode *counts, uint32_t *blen) {
    unsigned sym;
    int lk = pcode_lookup(counts, &d->br, &sym);
    if (lk != LK_OK) return BRO_UNEXPECTED_EOF;
    if (sym > 25) return BRO_INVALID_BLOCK_COUNT_CODE;
    uint32_t extra;
    if (br_bits(&d->br, BLEN_EXTRA[sym], &extra)) return BRO_UNEXPECTED_EOF;
    *blen = BLEN_BASE[sym] + extra;
    return 0;
}

/* parse_block_switch_command, src/lib.rs:1226-1250, plus the caller's bookkeeping
 * (:1187-1193, :1296-1302, :1379-1385). */
static int block_switch(Dec *d, BlockCat *c) {
    unsigned code;
    int lk = pcode_lookup(&c->types, &d->br, &code);
    if (lk == LK_NONE) return BRO_INVALID_BLOCK_SWITCH_COMMAND_CODE;
    if (lk == LK_EOF) return BRO_UNEXPECTED_EOF;
    unsigned nt = (code == 0) ? c->btype_prev : (code == 1) ? (c->btype + 1) % c->nbl : code - 2;
    uint32_t cnt;
    int rc = parse_block_count(d, &c->counts, &cnt);
    if (rc) return rc;
    c->btype_prev = c->btype;
    c->btype = nt;
    c->blen = cnt - 1;
    d->st.block_switches++;
    return 0;
}

/* One symbol of a block category: None -> nothing; Some(0) -> switch; Some(n) -> n-1
 * (src/lib.rs:1182-1197, 1294-1306, 1377-1389). */
static inline int block_tick(Dec *d, BlockCat *c) {
    if (c->blen == BLEN_INF) return 0;
    if (c->blen == 0) return block_switch(d, c);
    c->blen--;
    return 0;
}

/* NBLTYPESx + its two prefix codes + first block count: states NBltypesL..FirstBlockCountDistances,
 * src/lib.rs:1745-1885. */
static int parse_block_cat(Dec *d, BlockCat *c) {
    memset(c, 0, sizeof *c);
    c->btype = 0;
    c->btype_prev = 1;
    c->blen = BLEN_INF;
    int rc = parse_n_bltypes(d, &c->nbl);
    if (rc) return rc;
    if (c->nbl >= 2) {
        c->have_codes = 1;
        rc = parse_prefix_code(d, c->nbl + 2, &c->types);
        if (rc) return rc;
        rc = parse_prefix_code(d, 26, &c->counts);
        if (rc) return rc;
        rc = parse_block_count(d, &c->counts, &c->blen);
        if (rc) return rc;
    }
    return 0;
}

/* parse_context_map, src/lib.rs:1070-1144. */
static int parse_context_map(Dec *d, unsigned ntrees, uint8_t *cmap, size_t len) {
    unsigned rlemax = 0;
    int b = br_bit(&d->br);
    if (b < 0) return BRO_UNEXPECTED_EOF;
    if (b) {
        uint32_t v;
        if (br_bits(&d->br, 4, &v)) return BRO_UNEXPECTED_EOF;
        rlemax = v + 1;
    }
    PCode pc;
    int rc = parse_prefix_code(d, rlemax + ntrees, &pc);
    if (rc) {
        pcode_free(&pc);
        return rc;
    }
    size_t pushed = 0;
    while (pushed < len) {
        unsigned code;
        int lk = pcode_lookup(&pc, &d->br, &code);
        if (lk == LK_NONE) { rc = BRO_PARSE_ERROR_CONTEXT_MAP; break; }
        if (lk == LK_EOF) { rc = BRO_UNEXPECTED_EOF; break; }
        if (code > 0 && code <= rlemax) {
            uint32_t extra;
            if (br_bits(&d->br, code, &extra)) { rc = BRO_UNEXPECTED_EOF; break; }
            size_t repeat = ((size_t)1 << code) + extra;
            if (pushed + repeat > len) { rc = BRO_RUN_LENGTH_EXCEEDED_SIZE_OF_CONTEXT_MAP; break; }
            memset(cmap + pushed, 0, repeat);
            pushed += repeat;
        } else {
            cmap[pushed++] = (uint8_t)(code == 0 ? 0 : code - rlemax);
        }
    }
    pcode_free(&pc);
    if (rc) return rc;
    b = br_bit(&d->br);
    if (b < 0) return BRO_UNEXPECTED_EOF;
    if (b) bro_inverse_mtf(cmap, len);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Output                                                                     */
/* ------------------------------------------------------------------------- */
static inline int out_room(Dec *d, size_t n) {
    if (d->pos + n > d->cap) {
        d->needed = d->pos + n;
        return BRO_OUTPUT_TOO_SMALL;
    }
    return 0;
}

/* literal_buf (src/lib.rs:389, 407) starts as [0, 0] and receives every output byte -- literals
 * (:1361), copied / dictionary bytes (:2117), uncompressed bytes (:1726) -- so it always equals the
 * last two bytes of the output, zero padded at the start of the stream. */
static inline void sync_ctx(Dec *d) {
    d->p1 = d->pos >= 1 ? d->out[d->pos - 1] : 0;
    d->p2 = d->pos >= 2 ? d->out[d->pos - 2] : 0;
}

/* ------------------------------------------------------------------------- */
/* One compressed meta-block: header (states NBltypesL..PrefixCodesDistances, */
/* src/lib.rs:1745-2002) and the command loop (DataMetaBlockBegin..           */
/* CopyLiterals, src/lib.rs:2003-2141).                                       */
/* ------------------------------------------------------------------------- */
static int compressed_meta_block(Dec *d, size_t mlen) {
    BlockCat L, I, D;
    PCode *lit = NULL, *iac = NULL, *dst = NULL;
    unsigned ntrees_l = 0, ntrees_d = 0, n_iac = 0;
    uint8_t *cmap_l = NULL, *cmap_d = NULL;
    uint8_t cmode[256];
    int rc;
    memset(&L, 0, sizeof L);
    memset(&I, 0, sizeof I);
    memset(&D, 0, sizeof D);

    if ((rc = parse_block_cat(d, &L))) goto out;
    if ((rc = parse_block_cat(d, &I))) goto out;
    if ((rc = parse_block_cat(d, &D))) goto out;

    uint32_t v;
    if (br_bits(&d->br, 2, &v)) { rc = BRO_UNEXPECTED_EOF; goto out; } /* parse_n_postfix :548 */
    unsigned npostfix = v;
    if (br_bits(&d->br, 4, &v)) { rc = BRO_UNEXPECTED_EOF; goto out; } /* parse_n_direct :555 */
    unsigned ndirect = v << npostfix;
    for (unsigned i = 0; i < L.nbl; i++) { /* parse_context_modes_literals :562 */
        if (br_bits(&d->br, 2, &v)) { rc = BRO_UNEXPECTED_EOF; goto out; }
        cmode[i] = (uint8_t)v;
    }
    if ((rc = parse_n_bltypes(d, &ntrees_l))) goto out; /* parse_n_trees_l :575 */
    if (g_trace) fprintf(stderr, "NTL %u %u %zu\n", ntrees_l, L.nbl, mlen); /* analysis aid: literal trees, literal block types, MLEN */
    cmap_l = (uint8_t *)calloc(64 * (size_t)L.nbl, 1);
    if (!cmap_l) { rc = -1; goto out; }
    if (ntrees_l >= 2 && (rc = parse_context_map(d, ntrees_l, cmap_l, 64 * (size_t)L.nbl))) goto out;
    if ((rc = parse_n_bltypes(d, &ntrees_d))) goto out; /* parse_n_trees_d :582 */
    if (g_trace) fprintf(stderr, "NTD %u %u %u\n", ntrees_d, D.nbl, I.nbl); /* analysis aid: distance trees, distance / insert&copy block types */
    cmap_d = (uint8_t *)calloc(4 * (size_t)D.nbl, 1);
    if (!cmap_d) { rc = -1; goto out; }
    if (ntrees_d >= 2 && (rc = parse_context_map(d, ntrees_d, cmap_d, 4 * (size_t)D.nbl))) goto out;

    lit = (PCode *)calloc(ntrees_l, sizeof(PCode));
    iac = (PCode *)calloc(I.nbl, sizeof(PCode));
    dst = (PCode *)calloc(ntrees_d, sizeof(PCode));
    if (!lit || !iac || !dst) { rc = -1; goto out; }
    for (unsigned i = 0; i < ntrees_l; i++) /* parse_prefix_codes_literals :1016 */
        if ((rc = parse_prefix_code(d, 256, &lit[i]))) goto out;
    n_iac = I.nbl;
    for (unsigned i = 0; i < n_iac; i++) /* parse_prefix_codes_insert_and_copy_lengths :1034 */
        if ((rc = parse_prefix_code(d, 704, &iac[i]))) goto out;
    unsigned dist_alphabet = 16 + ndirect + (48u << npostfix);
    for (unsigned i = 0; i < ntrees_d; i++) /* parse_prefix_codes_distances :1052 */
        if ((rc = parse_prefix_code(d, dist_alphabet, &dst[i]))) goto out;

    const uint8_t *LUT0 = BRO_CONTEXT_LUT, *LUT1 = BRO_CONTEXT_LUT + 256, *LUT2 = BRO_CONTEXT_LUT + 512;
    size_t mb_count = 0; /* MetaBlock.count_output */

    for (;;) {
        /* parse_insert_and_copy_length :1179-1208 */
        const uint64_t tr_bit0 = d->br.pos; /* (trace only) */
        const size_t tr_pos0 = d->pos;
        if ((rc = block_tick(d, &I))) goto out;
        unsigned sym;
        int lk = pcode_lookup(&iac[I.btype], &d->br, &sym);
        if (lk == LK_NONE) { rc = BRO_PARSE_ERROR_INSERT_AND_COPY_LENGTH; goto out; }
        if (lk == LK_EOF) { rc = BRO_UNEXPECTED_EOF; goto out; }
        d->st.commands++;
        int implicit_zero = sym < 128; /* :2012-2015 */
        /* decode_insert_and_copy_length :1210-1224 */
        uint32_t ib, ie, cb, ce, extra;
        bro_insert_copy_entry(sym, &ib, &ie, &cb, &ce);
        if (br_bits(&d->br, ie, &extra)) { rc = BRO_UNEXPECTED_EOF; goto out; }
        size_t insert_len = ib + extra;
        if (br_bits(&d->br, ce, &extra)) { rc = BRO_UNEXPECTED_EOF; goto out; }
        size_t copy_len = cb + extra;
        if (mlen < mb_count + insert_len) { rc = BRO_EXCEEDED_EXPECTED_BYTES; goto out; } /* :2036 (Q4) */
        if ((rc = out_room(d, insert_len))) goto out;
        const uint64_t tr_bit1 = d->br.pos;

        /* parse_insert_literals :1286-1365 and the InsertLiterals state :2048-2081 */
        for (size_t k = 0; k < insert_len; k++) {
            if ((rc = block_tick(d, &L))) goto out;
            unsigned bt = L.btype;
            unsigned cid;
            switch (cmode[bt]) {
            case 0: cid = d->p1 & 0x3f; break;
            case 1: cid = d->p1 >> 2; break;
            case 2: cid = LUT0[d->p1] | LUT1[d->p2]; break;
            default: cid = (unsigned)(LUT2[d->p1] << 3) | LUT2[d->p2]; break;
            }
            unsigned idx = cmap_l[bt * 64 + cid];
            unsigned s;
            const uint64_t tr_lb = d->br.pos; /* (trace only) */
            lk = pcode_lookup(&lit[idx], &d->br, &s);
            if (g_trace > 1) /* analysis aid (BRO_TRACE=2): block type, context id, tree, code length of every literal */
                fprintf(stderr, "LIT %u %u %u %u\n", bt, cid, idx, (unsigned)(d->br.pos - tr_lb));
            if (lk == LK_NONE) { rc = BRO_PARSE_ERROR_INSERT_LITERALS; goto out; }
            if (lk == LK_EOF) { rc = BRO_UNEXPECTED_EOF; goto out; }
            d->out[d->pos + k] = (uint8_t)s;
            d->p2 = d->p1;
            d->p1 = (uint8_t)s;
        }
        d->pos += insert_len;
        mb_count += insert_len;
        d->st.literals += insert_len;
        if (g_trace && mb_count == mlen)
            fprintf(stderr, "CMDX %zu %llu %zu %zu %llu %llu\n", tr_pos0, (unsigned long long)tr_bit0, insert_len,
                    (size_t)copy_len, (unsigned long long)tr_bit1, (unsigned long long)d->br.pos);
        if (mb_count == mlen) break; /* :2069: copy part of the last command is ignored */
        const uint64_t tr_bit2 = d->br.pos;

        /* parse_distance_code :1367-1410 */
        unsigned dcode;
        if (implicit_zero) {
            dcode = 0;
        } else {
            if ((rc = block_tick(d, &D))) goto out;
            unsigned cid = copy_len >= 5 ? 3 : (unsigned)copy_len - 2;
            unsigned idx = cmap_d[D.btype * 4 + cid];
            lk = pcode_lookup(&dst[idx], &d->br, &dcode);
            if (lk == LK_NONE) { rc = BRO_PARSE_ERROR_DISTANCE_CODE; goto out; }
            if (lk == LK_EOF) { rc = BRO_UNEXPECTED_EOF; goto out; }
        }
        /* decode_distance :1412-1481 */
        uint64_t distance;
        if (dcode <= 3) {
            distance = d->dist[dcode];
        } else if (dcode <= 15) {
            int64_t base = (dcode <= 9) ? d->dist[0] : d->dist[1];
            int64_t sign = 2 * (int64_t)(dcode % 2) - 1;
            int64_t delta = (dcode <= 9) ? (dcode - 2) >> 1 : (dcode - 8) >> 1;
            int64_t r = base + sign * delta;
            if (r <= 0) { rc = BRO_INVALID_NON_POSITIVE_DISTANCE; goto out; }
            distance = (uint64_t)r;
        } else if (dcode <= 15 + ndirect) {
            distance = dcode - 15;
        } else {
            unsigned x = dcode - ndirect - 16;
            unsigned ndistbits = 1 + (x >> (npostfix + 1));
            if (br_bits(&d->br, ndistbits, &extra)) { rc = BRO_UNEXPECTED_EOF; goto out; }
            unsigned hcode = x >> npostfix;
            unsigned lcode = x & ((1u << npostfix) - 1);
            uint32_t offset = ((2 + (hcode & 1)) << ndistbits) - 4;
            distance = (uint32_t)(((offset + extra) << npostfix) + lcode + ndirect + 1);
        }
        size_t max_allowed = d->pos < d->window ? d->pos : d->window;
        if (dcode > 0 && distance <= max_allowed) { /* :1476-1478 */
            d->dist[3] = d->dist[2];
            d->dist[2] = d->dist[1];
            d->dist[1] = d->dist[0];
            d->dist[0] = (uint32_t)distance;
        }

        if (g_trace) { /* analysis aid (BRO_TRACE=1): one line per command on stderr */
            fprintf(stderr, "CMD %zu %zu %zu %llu\n", d->pos, insert_len, (size_t)copy_len,
                    distance <= max_allowed ? (unsigned long long)distance : 0ull);
            /* position / bit cursor at the command's start, after the insert&copy fields, after the literals, after
             * the distance; the raw distance and the last-distance ring after its update (tools/asm_emu.py) */
            fprintf(stderr, "CMDX %zu %llu %zu %zu %llu %llu %llu %u %llu %u %u %u %u\n", tr_pos0,
                    (unsigned long long)tr_bit0, insert_len, (size_t)copy_len, (unsigned long long)tr_bit1,
                    (unsigned long long)tr_bit2, (unsigned long long)d->br.pos, dcode, (unsigned long long)distance,
                    d->dist[0], d->dist[1], d->dist[2], d->dist[3]);
        }
        /* copy_literals :1483-1542 and the CopyLiterals state :2102-2141 */
        if (distance <= max_allowed) {
            if (mlen < mb_count + copy_len) { rc = BRO_EXCEEDED_EXPECTED_BYTES; goto out; } /* :2105 */
            if ((rc = out_room(d, copy_len))) goto out;
            uint8_t *dstp = d->out + d->pos;
            const uint8_t *srcp = dstp - distance;
            if (distance >= copy_len) {
                memcpy(dstp, srcp, copy_len);
            } else { /* window[i] = window[i % l], :1500-1503 */
                for (size_t k = 0; k < copy_len; k++) dstp[k] = srcp[k];
                d->st.overlapped_copies++;
            }
            d->pos += copy_len;
            mb_count += copy_len;
            d->st.copies++;
            d->st.copy_bytes += copy_len;
            if (distance > d->st.max_distance) d->st.max_distance = distance;
        } else {
            if (copy_len < 4 || copy_len > 24) { rc = BRO_INVALID_LENGTH_IN_STATIC_DICTIONARY; goto out; }
            uint64_t word_id = distance - max_allowed - 1;
            unsigned nbits = NDBITS[copy_len];
            uint64_t index = word_id & ((1u << nbits) - 1);
            uint64_t transform_id = word_id >> nbits;
            if (g_trace) fprintf(stderr, "TID %llu\n", (unsigned long long)transform_id);
            if (transform_id > 120) { rc = BRO_INVALID_TRANSFORM_ID; goto out; }
            uint8_t word[40];
            int wl = bro_transform((unsigned)transform_id, BRO_DICT + DOFFSET[copy_len] + index * copy_len,
                                   (unsigned)copy_len, word);
            if (wl < 0) { rc = BRO_REF_PANIC; goto out; }
            if (mlen < mb_count + (size_t)wl) { rc = BRO_EXCEEDED_EXPECTED_BYTES; goto out; } /* :2105 (Q4) */
            if ((rc = out_room(d, (size_t)wl))) goto out;
            memcpy(d->out + d->pos, word, (size_t)wl);
            d->pos += (size_t)wl;
            mb_count += (size_t)wl;
            d->st.dict_refs++;
            d->st.dict_bytes += (uint64_t)wl;
        }
        sync_ctx(d); /* every copied byte goes through literal_buf too (:2117) */
        if (mb_count == mlen) break; /* :2128 */
    }
    rc = 0;
out:
    if (L.have_codes) { pcode_free(&L.types); pcode_free(&L.counts); }
    if (I.have_codes) { pcode_free(&I.types); pcode_free(&I.counts); }
    if (D.have_codes) { pcode_free(&D.types); pcode_free(&D.counts); }
    if (lit) for (unsigned i = 0; i < ntrees_l; i++) pcode_free(&lit[i]);
    if (iac) for (unsigned i = 0; i < n_iac; i++) pcode_free(&iac[i]);
    if (dst) for (unsigned i = 0; i < ntrees_d; i++) pcode_free(&dst[i]);
    free(lit);
    free(iac);
    free(dst);
    free(cmap_l);
    free(cmap_d);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* Stream level: decompress(), src/lib.rs:1545-2170.                          */
/* ------------------------------------------------------------------------- */
int bro_decode(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *out_len, unsigned flags,
               bro_stats *stats) {
    bro_init();
    g_trace = getenv("BRO_TRACE") ? atoi(getenv("BRO_TRACE")) : 0;
    if (getenv("BRO_TRACE") && !g_trace) g_trace = 1; /* read once per stream: keeps the command loop free of libc calls */
    Dec d;
    memset(&d, 0, sizeof d);
    d.br.p = in;
    d.br.nbits = (uint64_t)in_len * 8;
    d.out = out;
    d.cap = out_cap;
    d.flags = flags;
    d.dist[0] = 4; /* RingBuffer::from_vec(vec![4, 11, 15, 16]): nth(0) == 4 */
    d.dist[1] = 11;
    d.dist[2] = 15;
    d.dist[3] = 16;
    int rc;
    unsigned wbits;
    uint32_t v;

    if ((rc = parse_wbits(&d, &wbits))) goto fin;
    d.window = ((size_t)1 << wbits) - 16;

    for (;;) { /* HeaderMetaBlockBegin :1572 */
        int is_last = br_bit(&d.br); /* parse_is_last :420 */
        if (is_last < 0) { rc = BRO_UNEXPECTED_EOF; goto fin; }
        if (is_last) {
            int empty = br_bit(&d.br); /* parse_is_last_empty :427 */
            if (empty < 0) { rc = BRO_UNEXPECTED_EOF; goto fin; }
            if (empty) break;
        }
        if (br_bits(&d.br, 2, &v)) { rc = BRO_UNEXPECTED_EOF; goto fin; } /* parse_m_nibbles :434 */
        unsigned mnibbles = (v == 3) ? 0 : v + 4;
        if (mnibbles == 0) { /* metadata block, accepted even when ISLAST (Q9); states :1617-1683 */
            int reserved = br_bit(&d.br);
            if (reserved < 0) { rc = BRO_UNEXPECTED_EOF; goto fin; }
            if (reserved) { rc = BRO_NON_ZERO_RESERVED_BIT; goto fin; }
            if (br_bits(&d.br, 2, &v)) { rc = BRO_UNEXPECTED_EOF; goto fin; } /* parse_m_skip_bytes :442 */
            unsigned mskipbytes = v;
            if (mskipbytes == 0) {
                if (br_byte_tail(&d.br) != 0) { rc = BRO_NON_ZERO_FILL_BIT; goto fin; }
            } else {
                /* parse_m_skip_len :449-467.  Q2: bytes are combined as byte << i, not << 8*i.
                 * Q10: the caller maps every error of this function, InvalidMSkipLen included, to
                 * UnexpectedEOF (:1661-1664). */
                uint32_t skip = 0, last = 0;
                for (unsigned i = 0; i < mskipbytes; i++) {
                    if (br_bits(&d.br, 8, &last)) { rc = BRO_UNEXPECTED_EOF; goto fin; }
                    skip |= last << i;
                }
                if (mskipbytes > 1 && last == 0) { rc = BRO_UNEXPECTED_EOF; goto fin; }
                skip += 1;
                if (br_byte_tail(&d.br) != 0) { rc = BRO_NON_ZERO_FILL_BIT; goto fin; }
                if (d.br.pos + 8ull * skip > d.br.nbits) { rc = BRO_UNEXPECTED_EOF; goto fin; }
                d.br.pos += 8ull * skip;
            }
        } else {
            /* parse_m_len :469-483 (read_u32_from_n_nibbles) */
            if (br_bits(&d.br, 4 * mnibbles, &v)) { rc = BRO_UNEXPECTED_EOF; goto fin; }
            if (mnibbles > 4 && (v >> ((mnibbles - 1) * 4)) == 0) { rc = BRO_NON_ZERO_TRAILER_NIBBLE; goto fin; }
            size_t mlen = (size_t)v + 1;
            int uncompressed = 0;
            if (!is_last) { /* :1689-1699 */
                uncompressed = br_bit(&d.br); /* parse_is_uncompressed :485 */
                if (uncompressed < 0) { rc = BRO_UNEXPECTED_EOF; goto fin; }
            }
            d.st.meta_blocks++;
            if (uncompressed) { /* :1701-1734 */
                if (br_byte_tail(&d.br) != 0) { rc = BRO_NON_ZERO_FILL_BIT; goto fin; }
                if (d.br.pos + 8ull * mlen > d.br.nbits) { rc = BRO_UNEXPECTED_EOF; goto fin; }
                if ((rc = out_room(&d, mlen))) goto fin;
                memcpy(d.out + d.pos, in + (d.br.pos >> 3), mlen);
                d.br.pos += 8ull * mlen;
                d.pos += mlen;
                d.st.raw_bytes += mlen;
                sync_ctx(&d); /* literal_buf.push per byte, :1726 */
            } else {
                rc = compressed_meta_block(&d, mlen);
                /* :1689-1693: with ISLAST set, an error of parse_n_bltypes_l is mapped to UnexpectedEOF --
                 * it can only be UnexpectedEOF anyway. */
                if (rc) goto fin;
            }
        }
        if (is_last) break; /* MetaBlockEnd :2146-2153 */
    }
    /* StreamEnd :2155-2167 */
    if (br_byte_tail(&d.br) != 0) { rc = BRO_NON_ZERO_TRAILER_BIT; goto fin; }
    if (d.br.pos < d.br.nbits) { rc = BRO_EXPECTED_END_OF_STREAM; goto fin; }
    rc = BRO_OK;
fin:
    if (rc < 0) rc = BRO_OUTPUT_TOO_SMALL; /* allocation failure: report as a capacity problem */
    *out_len = (rc == BRO_OUTPUT_TOO_SMALL && d.needed) ? d.needed : d.pos;
    d.st.bits_consumed = d.br.pos;
    if (stats) *stats = d.st;
    return rc;
}
