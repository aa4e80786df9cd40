/*
 * brx.h -- C ABI of the MI355X-native batched Brotli decompressor (libbrx.so).
 *
 * This is the drop-in boundary for the ONE hot path of ende76/brotli-rs: everything behind
 *     pub struct Decompressor<R: Read>            (reference src/lib.rs:377-394)
 *     pub fn new(r: R) -> Decompressor<R>          (reference src/lib.rs:398-410)
 *     impl<R: Read> Read for Decompressor<R>::read (reference src/lib.rs:2173-2193)
 * i.e. the private decompress() state machine (src/lib.rs:1545-2170) and the primitives under it
 * (src/bitreader, src/huffman, src/ringbuffer, src/lookuptable, src/dictionary, src/transformation).
 * The reference has no FFI of its own (pure safe Rust, `#![deny(unsafe_code)]`, src/lib.rs:1); the entry
 * points below are what a Rust `extern "C"` block for this path binds -- INTEGRATION.md shows the shim.
 *
 * Plain pointers and sizes only; no torch / HIP types in the signatures (a hipStream_t travels as void*).
 * Results are bit-exact with the reference: identical output bytes for valid streams and the identical
 * error kind (status 1..24 = DecompressorError in declaration order, src/lib.rs:294-319) for invalid ones.
 */
#ifndef BRX_H
#define BRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-stream status codes ------------------------------------------------------------------------
 * 0 = stream decoded; 1..24 = reference DecompressorError (src/lib.rs:294-319, same order);
 * 25/26 are additions of this build. */
enum {
    BRX_OK = 0,
    BRX_CODE_LENGTHS_CHECKSUM = 1,
    BRX_EXPECTED_END_OF_STREAM = 2,
    BRX_EXCEEDED_EXPECTED_BYTES = 3,
    BRX_INVALID_BLOCK_COUNT_CODE = 4,
    BRX_INVALID_BLOCK_SWITCH_COMMAND_CODE = 5,
    BRX_INVALID_LENGTH_IN_STATIC_DICTIONARY = 6,
    BRX_INVALID_MSKIP_LEN = 7, /* never observable, exactly like the reference (src/lib.rs:1661-1664) */
    BRX_INVALID_SYMBOL = 8,
    BRX_INVALID_TRANSFORM_ID = 9,
    BRX_INVALID_NON_POSITIVE_DISTANCE = 10,
    BRX_LESS_THAN_TWO_NON_ZERO_CODE_LENGTHS = 11,
    BRX_NO_CODE_LENGTH = 12,
    BRX_NON_ZERO_FILL_BIT = 13,
    BRX_NON_ZERO_RESERVED_BIT = 14,
    BRX_NON_ZERO_TRAILER_BIT = 15,
    BRX_NON_ZERO_TRAILER_NIBBLE = 16,
    BRX_PARSE_ERROR_CONTEXT_MAP = 17,
    BRX_PARSE_ERROR_COMPLEX_PREFIX_CODE_LENGTHS = 18,
    BRX_PARSE_ERROR_DISTANCE_CODE = 19,
    BRX_PARSE_ERROR_INSERT_AND_COPY_LENGTH = 20,
    BRX_PARSE_ERROR_INSERT_LITERALS = 21,
    BRX_RING_BUFFER_ERROR = 22,
    BRX_RUN_LENGTH_EXCEEDED_SIZE_OF_CONTEXT_MAP = 23,
    BRX_UNEXPECTED_EOF = 24,
    BRX_OUTPUT_TOO_SMALL = 25, /* capacity out_off[i+1]-out_off[i] exhausted; out_len[i] = bytes needed so far */
    BRX_REF_PANIC = 26,        /* the reference would panic here: UppercaseFirst on a dictionary word that
                                  starts with 0x00 (src/transformation/mod.rs:52-82) */
    BRX_INTERNAL_WATCHDOG = 27 /* bug guard inside the kernel.  Unreachable from any stream the reference terminates on, by construction:
                                  (1) the loop guard counts commands and meta-blocks against  8 * input bytes + capacity + 65536 --
                                  every command consumes a bit or emits a byte (one that does neither -- one-symbol codes throughout and a
                                  transform that leaves nothing of its word -- repeats for ever in the reference too, src/lib.rs:2003-2141);
                                  (2) the wait for a spill slab: since round 6 the pool holds one slab per wave of ALL launches in flight
                                  on the context (at most the 16 waves per CU the chip can hold at a time), so no wave ever waits --
                                  until then a second overlapping launch could starve a wave for 4 s and end a VALID stream with 27 */
};

/* ---- library-level return codes (not per-stream) ---------------------------------------------------- */
enum {
    BRX_SUCCESS = 0,
    BRX_ERR_INVALID_ARGUMENT = -1,
    BRX_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime failure at init: there is NO CPU fallback */
    BRX_ERR_HIP = -3,         /* a HIP call failed; brx_last_error() has the text */
    BRX_ERR_OUT_OF_MEMORY = -4
};

/* ---- options -------------------------------------------------------------------------------------- */
#define BRX_MEM_HOST 0u   /* every pointer argument is host memory; the library stages through HBM */
#define BRX_MEM_DEVICE 1u /* every pointer argument (data, offset tables, out_len, status) is device memory */
#define BRX_OPT_TIMING 2u /* record HIP-event timings of the kernels of this call (brx_last_timing) */
#define BRX_GEN_SWITCHES 8u /* brx_generate_batch only: two literal block types taking turns every 100 literals */
#define BRX_GEN_ADAPTIVE 16u /* brx_generate_batch only: the adaptive generator -- one wavefront per stream, prefix codes built
                                from each meta-block's own statistics (<= 15 bits, complex form with zero runs), two literal
                                trees behind a context map (mode UTF8), two literal block types switching every 1200 / 700
                                literals, last-distance codes; a slot of  len + len / 8 + 2048 * (len / metablock_bytes + 2)
                                bytes always suffices */
#define BRX_OPT_ORDER 4u  /* BRX_MEM_DEVICE only: queue the longest compressed streams first (a ragged batch finishes when
                             its longest stream does).  Costs one synchronous read of the offset table; the host-pointer
                             path always orders.  The work queue itself is dynamic: a wave that finishes a stream takes the
                             next one, so short streams never wait behind a fixed assignment. */

typedef struct brx_opts {
    uint32_t flags;   /* BRX_MEM_* | BRX_OPT_* */
    uint32_t reserved;
    void *hip_stream; /* hipStream_t to launch on, NULL = the context's own stream */
} brx_opts;

typedef struct brx_ctx brx_ctx; /* one per (process, GPU): device tables, spill-slab pool, HIP streams */

/* Threading and ordering contract.
 *  - Every entry point may be called from any thread.  Calls on ONE context are serialised on the host by a mutex
 *    (a batch call holds it until its work is enqueued -- and, for host pointers, finished); use one context per
 *    thread for host-side parallelism.  The reference Decompressor owns all of its state (src/lib.rs:377-394); a
 *    brx_stream does too once decoded.
 *  - Launches never share device state that matters: each gets its own work counter, and spill slabs are claimed by
 *    the waves themselves from a pool, so BRX_MEM_DEVICE calls enqueued on different HIP streams may overlap on the
 *    device.
 *  - BRX_MEM_DEVICE with hip_stream == NULL runs on the context's own NON-BLOCKING stream: it is not ordered after
 *    work on the caller's streams (not even the NULL stream).  Either pass the stream that produced the buffers in
 *    opts->hip_stream, or synchronise before the call.
 *  - A context that met streams whose prefix-code tables exceed the regular kernel's LDS also launches the wider kernel for
 *    them on a second HIP stream of its own, next to the regular kernel (forked from and joined back into the stream of the call
 *    with events: the call's stream order is unchanged, work enqueued behind the call sees all of its results).
 *  - Offset tables: in_off / out_off must be non-decreasing.  Host tables are checked (BRX_ERR_INVALID_ARGUMENT);
 *    in device memory a decreasing pair gives that stream an empty input (status 24) or zero capacity (status 25).
 *  - Per-stream limits of the 32-bit position arithmetic: output < 4 GiB - 256 B (more reports status 25 however
 *    large the capacity), input < 256 MiB for the fast path. */

/* Create a decoder context on HIP device `device` (0-based).  Fails with BRX_ERR_NO_DEVICE when no GPU is
 * present -- the product path never falls back to a CPU decoder. */
int brx_ctx_create(brx_ctx **out, int device);
void brx_ctx_destroy(brx_ctx *ctx);

/* Tuning and A/B knobs of a context: explicit arguments -- the library reads NO environment variable.  Takes effect for the
 * launches enqueued after the call; unknown options return BRX_ERR_INVALID_ARGUMENT.  The defaults are the measured best; the
 * test suite uses these to reach every path (the C++-only command loops, both builds of the assembly loop, the wider kernel
 * instances behind / next to the regular one, the lean instance for short streams). */
enum {
    BRX_OPTION_COMMAND_LOOP = 1,  /* 0 = assembly loop with the C++ loop as its safety net (default); 8 = the C++ loop alone, whole
                                     meta-blocks; 7 = the C++ loop alone, re-entered after every command; 6 = the default loop with every
                                     meta-block treated as one the assembly loop cannot take (also honoured by bounded readers made after it) */
    BRX_OPTION_LOOP_BUILD = 2,    /* -1 = by occupancy (default); 0 = bit window in VGPRs (full chip); 1 = in SGPRs (sparse launch) */
    BRX_OPTION_QUEUE_ORDER = 3,   /* 1 = longest compressed stream first on the host path (default); 0 = index order */
    BRX_OPTION_HAND_UP = 4,       /* 1 = streams whose tables spill a kernel's LDS go to the wider instance that holds them (default);
                                     0 = they stay, tables in an HBM slab */
    BRX_OPTION_LEVELS = 5,        /* how the wider instances are launched: 0 = one catch-all launch behind the regular kernel, 2 = a
                                     header-only classification pre-pass, then all four instances next to each other, each on
                                     its own list; 1 (default) = the latter for contexts that listed a stream within their last
                                     64 launches */
    BRX_OPTION_TINY_BYTES = 6,    /* compressed size up to which the regular kernel runs a stream in its C++ loop alone (128) */
    BRX_OPTION_HOST_IN_PLACE = 7, /* 1 = pinned host buffers are read / written by the kernel itself (default); 0 = staged copies */
    BRX_OPTION_GRID_CAP = 8,      /* 0 = none (default); else at most this many resident waves of the regular kernel */
    BRX_OPTION_SMALL_BYTES = 9,   /* compressed size up to which a stream goes to the lean instance first (default 128, at most
                                     500; 0 = no lean instance) */
    BRX_OPTION_SMALL_WAVES = 10,  /* waves per CU of the lean instance's grid (default 32) */
    BRX_OPTION_TRACE = 11,        /* 1 = every launch records when and where each stream was decoded (brx_last_trace); default 0 */
    BRX_OPTION_READER_WINDOW = 12, /* compressed bytes a bounded / pulled stream keeps resident on the device (default 8 MiB; 1 .. 256
                                     MiB): streams started afterwards */
    BRX_OPTION_LEVEL4 = 13,       /* 1 = behind every batch launch one more (usually empty, 4 us) launch of the level-4 instance -- 150 KiB
                                     of LDS, one per CU -- takes the streams whose prefix-code tables spill even level 3 (pieces of several
                                     MiB compressed in one go) (default); 0 = no such launch: those meta-blocks run in the C++ loop from a
                                     slab (~3 MB/s per stream).  For callers whose batches are tens of microseconds long and never hold
                                     such streams */
    BRX_OPTION_READER_MB_ROOM = 14 /* 1 = a bounded / pulled stream pauses in front of a compressed meta-block that does not fit behind its output
                                     window and lets the buffer grow to window + meta-block (at most ~34 MiB), so that the fast loop runs it
                                     (default); 0 = such meta-blocks decode command by command in the ~22 MiB buffer (~8 x slower): streams
                                     started afterwards */
};
int brx_ctx_set_option(brx_ctx *ctx, uint32_t option, int64_t value);

/* Decode `n` independent Brotli streams (the batch analogue of constructing n reference Decompressors and
 * calling read_to_end on each: benches/lib.rs:45-46, tests/lib.rs everywhere).
 *   in       concatenated compressed streams; stream i is in[in_off[i] .. in_off[i+1])
 *   in_off   n+1 offsets
 *   out      output arena; stream i may write out[out_off[i] .. out_off[i+1])  (capacity, not size)
 *   out_off  n+1 offsets
 *   out_len  n  decoded sizes (valid for status 0; "needed so far" for status 25).  For the other statuses: how far the decoder got --
 *               the slot's bytes [0, min(out_len, capacity)) are the stream's output in front of the error (the prefix a streaming
 *               reader has been handed; tools/prefix_fuzz.py holds it to the oracle); how far INTO the failing command that is, is
 *               not specified (nor by the reference: src/lib.rs:2173-2193 drops what the failing decompress() call had written)
 *   status   n  per-stream status codes (see above).  One bad stream never affects another.
 * Synchronous with respect to the host unless opts->hip_stream is given and BRX_MEM_DEVICE is set, in
 * which case the call only enqueues work on that stream -- EXCEPT under launch plan B (BRX_OPTION_LEVELS = 2, or a context that
 * met streams for the wider instances within its last 64 launches): the call then waits on the host for its classification
 * pre-pass (0.3 .. 1 ms: it reads six counters to size the instances' grids) before it enqueues the decode kernels and returns.
 * Everything it enqueues stays stream-ordered; only the host thread is held.  BRX_OPTION_LEVELS = 0 never waits.
 * Returns BRX_SUCCESS or a BRX_ERR_* code. */
int brx_decode_batch(brx_ctx *ctx, const uint8_t *in, const uint64_t *in_off, uint32_t n, uint8_t *out,
                     const uint64_t *out_off, uint64_t *out_len, int32_t *status, const brx_opts *opts);

/* The reference's exact error description strings (src/lib.rs:331-354, typos included) for 1..24. */
const char *brx_status_str(int32_t status);

/* Text of the last library-level error on this thread. */
const char *brx_last_error(void);

/* Timing of the most recent brx_decode_batch call made with BRX_OPT_TIMING (milliseconds, HIP events on
 * the launch stream).  which: 0 = whole device section, 1 = decode kernels. Returns <0 if unavailable.
 * which = 2 .. 7 (no BRX_OPT_TIMING needed; waits for the most recent launch): counters of that launch.  Streams whose
 * prefix-code tables do not fit the regular 6 912 B of LDS table memory are handed over on the device to the instance of the
 * kernel that holds them -- 9 472 / 17 152 / 37 632 B of table memory, 12 / 8 / 4 instead of 16 streams per CU:
 *   2, 3, 4  streams decoded at level >= 1, >= 2, 3 (2 = every stream that left the regular kernel)
 *   5        streams the lean instance (short streams, 32 per CU, launched in front of the regular kernel) left to the regular
 *            kernel -- the ones above its size limit plus the short ones it gave up on (any error, block switches, large tables)
 *   6        streams that were handed up at a LATER meta-block, with their decoder state (resumed there, not restarted)
 *   7        output bytes decoded twice because of hand-overs (0 = every such stream was resumed where it stood)
 *   8        (since the context was made) slices of bounded / pulled streams that paused in front of an item -- a header, an
 *            uncompressed block, a command -- that the RESIDENT input did not hold, to run it with more (brx_stream_new_reader)
 *   9        (since the context was made) pauses of bounded / pulled streams in front of ONE item (a long copy or insert, an uncompressed
 *            meta-block) that did not fit the room behind the output window: the window slides, and if that is not enough the buffer
 *            grows to hold the item
 *   10       streams of the most recent launch whose fast loop had read on past the end of the input (truncated / corrupted streams
 *            only; 0 for valid ones): they went back to a checkpoint a few dozen dwords in front of the end and were finished with
 *            the exact end-of-input rules (round 5 took the whole meta-block back: a cut 1 MiB stream stalled its batch)
 *   11       streams of the most recent launch that the level-3 kernels handed on to the level-4 instance (150 KiB of LDS, one per
 *            CU: meta-blocks with more than 37.6 KiB of prefix-code tables -- one piece of several MiB from an encoder)
 *   12       (since the context was made) waves that did not get a spill slab at their first pass over the pool.  0 by construction
 *            since round 6: the pool has a slab for every wave of every launch in flight on the context (BRX_INTERNAL_WATCHDOG)
 *   13       slabs of the context's spill pool (896 KiB each; it grows with the launches in flight, to at most what the chip runs at a time)
 *   14, 15   (since the context was made) batches the Read facade launched for queued streams / streams in them: how well the host's
 *            threads coalesce (brx_stream_new below; a status-25 retry counts again)
 */
double brx_last_timing(brx_ctx *ctx, int which);

/* Diagnostics (BRX_OPTION_TRACE = 1): 4 words per stream of the most recent launch -- start and end of its decode on the GPU's
 * 100 MHz realtime counter, HW_ID register (XCC / SE / CU / SIMD / wave slot of the wave that decoded it) | kernel level << 32,
 * workgroup index | grid size << 32.  Streams decoded by the lean instance have all-zero records.  Waits for that launch.
 * While the option is on, the host-pointer path decodes a batch in ONE launch (no chunks on separate HIP streams). */
int brx_last_trace(brx_ctx *ctx, uint64_t *dst, uint32_t n);

/* Blocks until everything enqueued on the context's stream (or `hip_stream`) has finished. */
int brx_synchronize(brx_ctx *ctx, void *hip_stream);

/* Pinned (page-locked, device-mapped) host memory for the host-pointer path.  With buffers from brx_host_alloc (or
 * hipHostMalloc / hipHostRegister with the mapped flag) brx_decode_batch makes no copies around the kernel: the kernel
 * reads the compressed bytes in place (a wave stages its input 256 bytes ahead of its cursor, the PCIe round trip hides
 * behind ~40 us of decoding) and stores every output byte to the host buffer itself while it decodes, next to the copy
 * in HBM that serves as the stream's window -- the device-to-host transfer rides on the decode as 1 KiB writes.  The
 * output pointer must sit at the 16-byte phase of its offsets (out + out_off[0] 16-byte aligned), else -- and with
 * pageable memory -- the batch is staged: input copy, decode, output copy, in chunks on their own HIP streams from the
 * second grid-full of streams on.  With the output stored in place only the stream's bytes are written; a staged copy
 * brings a slot's slack along.  (Ingest shape of the reference's file walker, src/main.rs:49-70: read files straight
 * into such a buffer, decode, write out -- brotli-rs_amd/host/brx_walk.cpp.) */
void *brx_host_alloc(size_t bytes);
void brx_host_free(void *p);

/* ---- Stream generator (the reference has no encoder, README.md:1) -------------------------------
 * Makes n valid Brotli streams ON THE DEVICE from n inputs: a minimal encoder, one GPU thread per stream -- greedy LZ77
 * over a 2048-entry hash table, the input cut into meta-blocks of `metablock_bytes` (0 = 65536, at most 2^24), every
 * meta-block with one block type per category and three static complete prefix codes sent in complex form (literals
 * 8 bits, insert&copy symbols 9/10 bits, distances 6 bits + extra; csrc/brx_gen.hip).  For batches of real, compressible
 * streams of any size without committed fixtures and for differential fuzzing: every stream it makes decodes back to its
 * input with brx_decode_batch and with any conforming decoder.
 *   src / src_off   n inputs, concatenated (input i is src[src_off[i] .. src_off[i+1]), may be empty)
 *   out / out_off   n output slots; a slot of  len + len / 8 + 256 * (len / metablock_bytes + 2)  bytes always suffices
 *   out_len         n  compressed sizes
 *   status          n  0, or 25 when the slot was too small (out_len then says how much was needed)
 * Pointers are host memory (staged) or, with BRX_MEM_DEVICE in opts->flags, device memory (opts->hip_stream as for
 * brx_decode_batch).  BRX_GEN_SWITCHES in opts->flags: every meta-block declares two literal block types (sharing the
 * one literal tree) and switches between them every 100 literals -- the block-switch commands of the format, 4 bits each.  Returns BRX_SUCCESS or a BRX_ERR_* code.
 * The generator's hash tables live in ONE scratch buffer per context: device-pointer calls on different HIP streams of one
 * context are not concurrent-safe (use one context per stream); a call that needs a larger scratch synchronizes the device
 * before it replaces the old one. */
int brx_generate_batch(brx_ctx *ctx, const uint8_t *src, const uint64_t *src_off, uint32_t n, uint8_t *out,
                       const uint64_t *out_off, uint64_t *out_len, int32_t *status, uint32_t metablock_bytes,
                       const brx_opts *opts);

/* ---- the decoded bytes of a batch, back to back (device memory only) ------------------------------------
 * brx_decode_batch leaves stream i in a slot sized for the worst case; whoever ships the results on (the ragged gather of
 * SURVEY 8e, a writer of one concatenated file) wants them without the slack:
 *   dst[dst_off[i] .. dst_off[i] + len[i])  =  out[out_off[i] .. out_off[i] + len[i])      for i < n
 * `len` is brx_decode_batch's out_len (zero the entries of failed streams first); dst_off is its exclusive prefix sum
 * (n entries, computed by the caller, e.g. torch.cumsum) and `total` its grand total.  All pointers are DEVICE memory.
 * hip_stream NULL = the context's own stream and the call returns when the copy is done; otherwise it is enqueued.
 * One pass at HBM rate (reads `total`, writes `total`); no reference counterpart (a `Decompressor` owns one stream). */
int brx_compact_batch(brx_ctx *ctx, const uint8_t *out, const uint64_t *out_off, const uint64_t *len, uint32_t n,
                      uint8_t *dst, const uint64_t *dst_off, uint64_t total, void *hip_stream);

/* ---- Read-shaped stream facade (one object = one stream, like one reference Decompressor) ----------
 * brx_stream_new copies the compressed bytes and queues the stream on its context.  The first brx_stream_read of
 * ANY queued stream decodes ALL streams queued on that context in one batch (N live Decompressors cost about one
 * batch, not N launches).  Threads coalesce by themselves (round 6): making a stream never waits for a running batch, so the streams
 * that other threads make while one batch runs go out together as the next one -- ONE reader leads it, the others wait for their
 * own stream only (T threads that each make a stream, read it and free it ride batches of T / 2 .. T -- the leading reader waits a moment, 120 us of
 * quiet and 1.5 ms at most, for the others while batches are small: 16 threads on alice29 324 MB/s, 64 threads 1.15 GB/s, 512 threads 2.9 GB/s, where turns of one stream each gave 24 MB/s; profiles/r06_stream_threads.txt; the context keeps the pinned staging of its largest facade batch, at most 1 GiB).  Later reads serve slices:  n>0 bytes read, 0 at end of stream forever after (reference
 * src/lib.rs:2155-2166).  For an invalid stream the bytes produced before the error are served first, then every
 * read returns -status (the reference returns io::ErrorKind::InvalidData carrying brx_status_str(status) after an
 * unspecified prefix, src/lib.rs:2177).  Values below -900 are library failures (-1000 + BRX_ERR_*),
 * e.g. a stream that expands past the 4 GiB - 256 B per-stream limit; brx_last_error() has the text. */
typedef struct brx_stream brx_stream;
brx_stream *brx_stream_new(brx_ctx *ctx, const uint8_t *in, size_t n);
/* The same object in BOUNDED mode (brx_stream_new picks it by itself for compressed inputs of 4 MiB and more): the stream
 * is decoded slice by slice -- about 4 MiB of output per brx_stream_read that runs dry -- by a resumable kernel into a
 * sliding window on the device, so the output resident at any time is bounded by the largest Brotli window (16 MiB) plus
 * one slice plus slack (~22 MiB) however large the stream (like the reference's Decompressor, whose state is its window,
 * src/lib.rs:377-394, 1560-1567).  Reads see decoded bytes as the slices complete; an invalid stream serves everything
 * decoded before the error.  A single command that produces more than the slack (a > 1 MiB copy, insert or uncompressed
 * meta-block) is taken back by the kernel, which pauses in front of it; the window slides and, if the command still does not fit, the
 * buffer grows to hold it (at most ~48 MiB more: brx_last_timing 9 counts such pauses);
 * only if that allocation fails does the stream fall back to whole-stream decoding. */
brx_stream *brx_stream_new_bounded(brx_ctx *ctx, const uint8_t *in, size_t n);
/* The bounded reader over a SOURCE instead of a buffer: the reference's Decompressor::new(r: R) with R: Read
 * (src/lib.rs:398-410), which pulls its input through a BufReader as it decodes (src/bitreader/mod.rs:21-53).  `read` is called
 * -- from inside brx_stream_read, on the caller's thread -- for up to `cap` more compressed bytes; it returns how many it gave,
 * 0 = end of input (and is not called again).  Compressed input is held in a sliding 8 MiB device window the same way the
 * output is (BRX_OPTION_READER_WINDOW): a stream of any length decodes with about 32 MiB of buffers on the device and 1 MiB on the
 * host.  A slice pauses before its resident input runs out -- in front of the header, uncompressed block or command that would
 * not fit -- and goes on once more is resident.  One item that needs more than the window holds (an uncompressed meta-block is
 * up to 16 MiB) makes the window grow, doubling up to 256 MiB; one command that produces more than the slack makes the output
 * buffer grow (see above).  `read` runs with the context's lock released: it may itself read from another brx_stream of the
 * same context (a Decompressor over a Decompressor).  A callback that returns more than `cap` fails the stream
 * (BRX_ERR_INVALID_ARGUMENT).  The context must outlive every read. */
typedef size_t (*brx_read_fn)(void *user, uint8_t *buf, size_t cap);
brx_stream *brx_stream_new_reader(brx_ctx *ctx, brx_read_fn read, void *user);
int64_t brx_stream_read(brx_stream *s, uint8_t *buf, size_t len);
void brx_stream_free(brx_stream *s);

/* ---- The node: the GPUs of one machine behind ONE call (round 6) ---------------------------------------------
 * Streams are independent -- a reference Decompressor owns all of its state (src/lib.rs:378-394) -- so a batch shards trivially: a
 * brx_node owns one brx_ctx per GPU (one process, one host thread per GPU), deals the streams of a batch over them, decodes the
 * shards at the same time and leaves every result where the caller said, exactly as brx_decode_batch would have.  This is how a
 * host that is not Python (the Rust `Decompressor` pool of INTEGRATION.md) reaches GPUs 1 .. 7.
 *   BRX_MEM_HOST    every GPU reads its streams from, and writes its results to, the caller's buffers: no exchange between GPUs at
 *                   all (pinned buffers -- brx_host_alloc -- are used in place by every GPU's kernel; pageable ones are staged per GPU).
 *   BRX_MEM_DEVICE  every pointer is memory of the ROOT rank's GPU.  The other ranks get exactly their shard of the compressed bytes
 *                   over xGMI (scatter), decode, compact their results and send them back (ragged gather); the root expands them
 *                   into the caller's slots.  One grouped exchange each way: RCCL (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd,
 *                   one communicator per GPU in this process; librccl is loaded the first time it is needed) or plain peer copies
 *                   (hipMemcpyPeerAsync) -- BRX_NODE_OPTION_TRANSPORT.  No collective ever runs inside the decode.
 * Dealing (brx_node_opts::deal):
 *   BRX_NODE_DEAL_RANGES  rank r of G takes the contiguous index range [r * n / G, (r + 1) * n / G)            (default)
 *   BRX_NODE_DEAL_BYTES   contiguous ranges cut at equal shares of the COMPRESSED bytes (a batch that is sorted or drifts in size)
 *   BRX_NODE_DEAL_SNAKE   the streams sorted by compressed size, largest first, and dealt 0 .. G-1, G-1 .. 0, ... with every rank
 *                         keeping the stream count of its index range: a ragged batch within a few percent of equal bytes AND equal
 *                         counts per GPU.  Not contiguous, so every rank's shard is packed and its results are unpacked (one more
 *                         pass over the data: host memcpy under BRX_MEM_HOST, an HBM-rate gather kernel under BRX_MEM_DEVICE).
 * How many GPUs a batch is dealt over (brx_node_opts::use_gpus = 0): one per BRX_NODE_OPTION_MIN_STREAMS streams (default: what one
 * GPU decodes at a time, 16 per CU = 4096 on MI355X) -- a stream is a serial job of one wavefront, and 512 of them take a GPU about as
 * long as 4096; give use_gpus (or lower the option) to trade GPUs for latency.
 * The devices of a node need not differ: several ranks on one GPU ("virtual ranks") run the same code -- contexts, threads,
 * dealing, scatter, gather -- which is how the path is tested on a one-GPU box.
 * Results, status codes and error behaviour per stream are those of brx_decode_batch.  The call returns when everything is done. */
typedef struct brx_node brx_node;

#define BRX_NODE_DEAL_RANGES 0u
#define BRX_NODE_DEAL_BYTES 1u
#define BRX_NODE_DEAL_SNAKE 2u

typedef struct brx_node_opts {
    uint32_t flags;   /* BRX_MEM_HOST | BRX_MEM_DEVICE | BRX_OPT_TIMING */
    uint32_t deal;    /* BRX_NODE_DEAL_* */
    int32_t use_gpus; /* 0 = the library picks (see above); else the batch is dealt over ranks 0 .. use_gpus - 1 (root included) */
    int32_t root;     /* BRX_MEM_DEVICE: the rank whose GPU holds the buffers (0 <= root < use_gpus) */
    void *hip_stream; /* BRX_MEM_DEVICE: a stream of the root's GPU the call's work is ordered behind (NULL: none) */
} brx_node_opts;

enum {
    BRX_NODE_OPTION_TRANSPORT = 100,    /* 0 = RCCL where every rank has a GPU of its own, peer copies otherwise (default); 1 = peer
                                           copies (hipMemcpyPeerAsync); 2 = RCCL (fails with BRX_ERR_INVALID_ARGUMENT for virtual ranks,
                                           BRX_ERR_HIP if librccl cannot be loaded) */
    BRX_NODE_OPTION_MIN_STREAMS = 101,  /* streams per GPU below which use_gpus = 0 does not add a GPU (0 = the default, see above) */
    BRX_NODE_OPTION_EXCHANGE_ROOT = 102 /* 1 = under BRX_MEM_DEVICE the root's own shard travels through the transport as well (to
                                           itself) instead of being decoded in place: exercises the send / receive pair on one GPU */
};

/* devices: n_devices HIP device indices, rank r on devices[r]; NULL / 0 = every visible GPU, one rank each.  Fails like
 * brx_ctx_create (BRX_ERR_NO_DEVICE without a GPU: there is no CPU fallback). */
int brx_node_create(brx_node **out, const int *devices, int n_devices);
void brx_node_destroy(brx_node *node); /* (not while a call on the node is in flight on another thread) */
int brx_node_size(const brx_node *node);
/* The context of a rank (owned by the node): for brx_ctx_set_option / brx_last_timing on one GPU.  Do not destroy it. */
brx_ctx *brx_node_ctx(brx_node *node, int rank);
/* BRX_OPTION_* : applied to every rank's context.  BRX_NODE_OPTION_* : the node's own. */
int brx_node_set_option(brx_node *node, uint32_t option, int64_t value);
/* brx_decode_batch over the node.  Arguments as there; see the block comment above for where the pointers live. */
int brx_node_decode_batch(brx_node *node, const uint8_t *in, const uint64_t *in_off, uint32_t n, uint8_t *out,
                          const uint64_t *out_off, uint64_t *out_len, int32_t *status, const brx_node_opts *opts);
/* The dealing of brx_node_decode_batch on its own (host arithmetic, no GPU is touched): order[k] = the caller's index of the k-th
 * stream in dealt order (the identity for the two contiguous deals), rank r takes order[cut[r] .. cut[r + 1]).  in_off: n + 1
 * offsets in host memory; order: n entries; cut: gpus + 1 entries.  For callers that place their data themselves, and for tests. */
int brx_node_deal(const uint64_t *in_off, uint32_t n, int gpus, uint32_t deal, uint32_t *order, uint32_t *cut);
/* Of the most recent brx_node_decode_batch.  which: 0 = ranks the batch was dealt over; 1 = streams of `rank`; 2 = compressed
 * bytes of `rank`; 3 = decode-kernel milliseconds of `rank` (BRX_OPT_TIMING, else -1); 4 = wall milliseconds of the whole call;
 * 5 = wall milliseconds until every shard was on its GPU (BRX_MEM_DEVICE: the scatter; else 0); 6 = 1 if the exchange went through
 * RCCL, 0 for peer copies / no exchange.  Returns < 0 if unavailable. */
double brx_node_last_timing(brx_node *node, int which, int rank);

#ifdef __cplusplus
}
#endif
#endif
