/*
 * brotli_oracle.h -- CPU restatement of ende76/brotli-rs's decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under brotli-rs_amd/ (the product) may
 * include, link or call this.  Allowed users: tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg -- as the checker / reported baseline, never
 * as the thing measured or shipped.
 *
 * Parity status: the reference is Rust (crate `brotli` v0.3.23, no dependencies)
 * and there is no Rust toolchain in the build image, so oracle/_ref does not
 * exist.  The restatement is pinned by the reference's own vectors instead:
 * all 43 valid data/ pairs, the 9 frewsxcv reject streams, every inline vector
 * of tests/lib.rs and the 121 transform unit vectors (tests/golden/).
 * Parity UNPINNED (no reference test reaches them; behaviour follows the cited
 * source lines only): Q1 OmitFirstN on short words, Q2 MSKIPLEN for
 * MSKIPBYTES>=2, Q3 UppercaseFirst on a 0x00-leading word, Q10, Q15 incomplete
 * complex prefix codes.  See DESIGN.md "Oracle".
 */
#ifndef BROTLI_ORACLE_H
#define BROTLI_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes: 0 OK; 1..24 = DecompressorError in declaration order
 * (reference src/lib.rs:294-319); 25/26 are this build's additions. */
enum {
    BRO_OK = 0,
    BRO_CODE_LENGTHS_CHECKSUM = 1,
    BRO_EXPECTED_END_OF_STREAM = 2,
    BRO_EXCEEDED_EXPECTED_BYTES = 3,
    BRO_INVALID_BLOCK_COUNT_CODE = 4,
    BRO_INVALID_BLOCK_SWITCH_COMMAND_CODE = 5,
    BRO_INVALID_LENGTH_IN_STATIC_DICTIONARY = 6,
    BRO_INVALID_MSKIP_LEN = 7,
    BRO_INVALID_SYMBOL = 8,
    BRO_INVALID_TRANSFORM_ID = 9,
    BRO_INVALID_NON_POSITIVE_DISTANCE = 10,
    BRO_LESS_THAN_TWO_NON_ZERO_CODE_LENGTHS = 11,
    BRO_NO_CODE_LENGTH = 12,
    BRO_NON_ZERO_FILL_BIT = 13,
    BRO_NON_ZERO_RESERVED_BIT = 14,
    BRO_NON_ZERO_TRAILER_BIT = 15,
    BRO_NON_ZERO_TRAILER_NIBBLE = 16,
    BRO_PARSE_ERROR_CONTEXT_MAP = 17,
    BRO_PARSE_ERROR_COMPLEX_PREFIX_CODE_LENGTHS = 18,
    BRO_PARSE_ERROR_DISTANCE_CODE = 19,
    BRO_PARSE_ERROR_INSERT_AND_COPY_LENGTH = 20,
    BRO_PARSE_ERROR_INSERT_LITERALS = 21,
    BRO_RING_BUFFER_ERROR = 22,
    BRO_RUN_LENGTH_EXCEEDED_SIZE_OF_CONTEXT_MAP = 23,
    BRO_UNEXPECTED_EOF = 24,
    BRO_OUTPUT_TOO_SMALL = 25, /* caller capacity exhausted; *out_len = bytes needed so far */
    BRO_REF_PANIC = 26         /* the reference would panic here (Q3: src/transformation/mod.rs:52-82) */
};

/* flags */
#define BRO_FLAG_TREE_WALK 1u /* structure-faithful prefix lookup: implicit heap array walked one
                                 bit per level exactly like src/huffman/tree/mod.rs:63-93 (slow);
                                 default is an equivalent canonical first-code decoder */

/* Per-stream census of what the stream exercises (SURVEY.md section 8d figures). */
typedef struct {
    uint64_t meta_blocks;      /* compressed or uncompressed meta-blocks with MLEN>0 */
    uint64_t commands;         /* insert&copy commands */
    uint64_t literals;         /* bytes produced by prefix-coded literals */
    uint64_t raw_bytes;        /* bytes of uncompressed meta-blocks */
    uint64_t copies;           /* window copies */
    uint64_t copy_bytes;       /* bytes produced by window copies */
    uint64_t overlapped_copies;/* window copies with distance < length */
    uint64_t dict_refs;        /* static dictionary references */
    uint64_t dict_bytes;       /* bytes produced by (transformed) dictionary words */
    uint64_t block_switches;   /* block-switch commands, all three categories */
    uint64_t bits_consumed;    /* input bits consumed */
    uint64_t max_distance;     /* largest window distance used */
} bro_stats;

/* Decode one whole stream.  `out` doubles as the sliding window.  Returns a status code.
 * On BRO_OK *out_len is the decoded size.  On an error status *out_len is the number of bytes
 * produced before the error (NOT a stable observable of the reference). */
int bro_decode(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *out_len,
               unsigned flags, bro_stats *stats /* may be NULL */);

/* The exact description strings of src/lib.rs:331-354 (typos included); 25/26 get this build's text. */
const char *bro_status_str(int status);

/* One word transformation, src/transformation/mod.rs:84-209 (+ Q1, Q3).  `out` needs 24+13 bytes.
 * Returns the transformed length, or -1 where the reference would panic (Q3). */
int bro_transform(unsigned id, const uint8_t *word, unsigned len, uint8_t *out);

/* Inverse move-to-front, src/lib.rs:1164-1177. */
void bro_inverse_mtf(uint8_t *v, size_t n);

/* Table access for tests. */
const uint8_t *bro_dictionary(void);              /* 122784 bytes */
const uint8_t *bro_context_lut(int which);        /* 0,1,2 -> 256 bytes */
void bro_insert_copy_entry(unsigned sym, uint32_t *ins_base, uint32_t *ins_extra, uint32_t *copy_base,
                           uint32_t *copy_extra); /* sym < 704 */

#ifdef __cplusplus
}
#endif
#endif
