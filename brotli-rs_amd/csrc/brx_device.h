// brx_device.h -- shared definitions between the host side (brx_api.cpp) and the gfx950 kernels.
#pragma once
#include <stdint.h>

// Geometry of one decoder wave.  One workgroup = one 64-lane wavefront = one stream at a time.
#define BRX_WAVE 64
#define BRX_RING_BYTES 2048u     // LDS sliding-window ring: last 2 KiB of the stream's output
// Four instances of the kernel (five with level 4, below), one source (brx_kernels.hip; brx_kernels_l1/l2/l3/l4.hip set BRX_LEVEL): the regular one with
// 10 KiB of LDS per wave (16 streams per CU) and three wider ones for streams whose meta-block tables do not fit its table
// memory (real text at high quality: lcet10.txt needs 2 208 words, 800 KB of text at quality 11 ~5 000):
//   level   LDS per wave   table memory          streams per CU
//     0       10 240 B      1 728 words            16
//     1       12 800 B      2 368 words            12
//     2       20 480 B      4 288 words             8
//     3       40 960 B      9 408 words             4
//     4      153 600 B     37 568 words             1      (round 5; fed by level 3 only)
// (sizes are multiples of 1 280 B so that the waves of a CU fill its 160 KiB whatever the allocation granule.)  A wave that
// finds a stream spilling its level's table memory drops it and lists it for the level that holds its tables
// (BrxKernelArgs::defer, no host round trip).  Beyond level 4: the spill slabs.
#ifndef BRX_LEVEL
#define BRX_LEVEL 0
#endif
#define BRX_LEVELS 5
#define BRX_LIST_REGIONS 6  // regions of one launch in brx_ctx::d_defer: lists 0..2, the late list, the lean kernel's list, the class bytes
#define BRX_LATE_CAP 16384u // entries of the late list (with state records); a stream that finds it full keeps its slab
#define BRX_LATE2_CAP 4096u // entries of the second late list (level 3 -> level 4)
#if BRX_LEVEL == 0
#define BRX_LDS_GROW 0u
#elif BRX_LEVEL == 1
#define BRX_LDS_GROW 2560u
#elif BRX_LEVEL == 2
#define BRX_LDS_GROW 10240u
#elif BRX_LEVEL == 3
#define BRX_LDS_GROW 30720u
#else
// Level 4 (round 5): ONE workgroup per CU, 150 KiB of LDS (gfx950 lets a workgroup declare all 160 KiB of the CU's): 37 568 words of
// table memory for the meta-blocks the reference encoder makes of one heterogeneous piece of more than a megabyte (80 .. 250 literal trees, dozens
// of block types: 10 .. 35 k words of tables; profiles/r05_big_trees.txt).  Launched behind the level-3 catch-all; its only input is the
// SECOND late list (BrxKernelArgs::handup2), which level-3 kernels fill.  Its Lds has the table memory LAST (brx_kernels.hip), so that
// everything else keeps an offset a DS instruction's 16-bit immediate can hold.
#define BRX_LDS_GROW 143360u
#endif
// A fifth, LEAN instance (brx_kernels_s.hip, BRX_SMALL): 5 120 B of LDS per wave and <= 64 VGPRs = 32 waves per CU, for
// streams of at most BRX_SMALL_STREAM_BYTES compressed bytes (a batch of short messages, the RLE-like fills of BASELINE
// configs 3 / 4): the whole input staged across the lanes once, a straight-line header for one block type per category, a
// command loop of its own, no assembly loop, no parked state.  It decodes what is valid and plain and LISTS everything
// else -- larger streams, any error, block switches, tables beyond its 512 words -- for the regular kernel behind it.
#ifdef BRX_SMALL
#define BRX_TM_WORDS 512u
#define BRX_LENS_BYTES 768u
#else
#define BRX_TM_WORDS (1728u + BRX_LDS_GROW / 4u) // LDS table memory (prefix-code tables, context maps): 6 912 B at level 0
#define BRX_LENS_BYTES 1280u     // LDS: code-length scratch (768 B) + parked decoder state (512 B)
#endif
#define BRX_SMALL_MAX_BYTES 500u    // what the lean instance can stage (+ up to 3 bytes of misalignment: 126 of its 128 dwords)
#define BRX_SMALL_STREAM_BYTES 128u // what it is given by default: its compiled command loop beats the regular kernel's hand-overs
                                    // on a handful of commands (ukkonooa, 69 B: 0.76 -> 0.26 ms per 16 384), not on the 89 commands
                                    // and 381 literals of monkey (425 B: 0.97 -> 1.74 ms) -- profiles/r04_lean.txt
#define BRX_TINY_STREAM_BYTES 128u // compressed streams up to this size run their commands in the C++ loop alone
#define BRX_FLUSH_BLOCK 1024u    // ring -> HBM flush granule: 64 lanes x 16 B, address aligned
#define BRX_FLUSH_LAG 0u         // a block is flushed once the write cursor is this far past its end (everything that
                                 // is still in flight lands before a flush, so no lag is needed)

// Spill slab in HBM for tables that do not fit the LDS table memory (worst case: 256 trees per category,
// SURVEY 2.2).  Sized for the worst case so a meta-block can never run out of table memory.  Slabs live in a
// pool; a wave claims one (atomic bitmap) the first time a stream spills and releases it when the stream ends,
// so concurrent launches on one context share the pool safely.
#define BRX_SCRATCH_WORDS (224u * 1024u) // 896 KiB per slab

struct BrxSlabPool {
    uint32_t *bitmap; // count / 32 words, bit set = slab in use
    uint32_t *slabs;  // count * BRX_SCRATCH_WORDS
    uint32_t count;   // multiple of 32
    uint32_t *sink;   // one slab outside the bitmap: where the tables of a wave go that could not claim one within 4 s (a pool
                      // that stays exhausted is a bug); its stream ends with the watchdog status, the device does not hang
    uint32_t *waits;  // one word: waves that did not get a slab at their first try (brx_last_timing 12).  Stays 0: the host sizes the
                      // pool for every wave of the launches in flight (brx_api.cpp, pool_need)
};

// One static-dictionary word transform (spec Appendix B): prefix + elementary op + suffix.
struct BrxTransform {
    uint8_t prefix[8];
    uint8_t suffix[8];
    uint8_t plen, slen, op, pad;
};

// Read-only device tables owned by a brx_ctx.
struct BrxDeviceTables {
    const uint8_t *dict;        // 122784 B, spec Appendix A
    const uint8_t *context_lut; // Lut0 | Lut1 | Lut2, 3 x 256 B
    const BrxTransform *xforms; // 121 entries
    const uint32_t *iac;        // insert&copy symbol records for the assembly loop (brx_hot.S): 704 x {insert base, copy
                                // base, 2 * distance context (4 = implicit distance 0), insert extra bits | copy extra bits
                                // << 8}, then at dword 2816 64 dwords DOFFSET | NDBITS << 24
};

// Resumable decode of ONE stream with bounded output memory (the Read facade for very large streams):
// the kernel stops at the first command boundary at or beyond `pause_at`, flushes its ring, parks the whole wave state
// (its LDS) here and reports BRX_PAUSED; the next launch picks up from it against a slid output window.
// The compressed input is a sliding window too (the reference pulls its input through a BufReader as it decodes,
// src/bitreader/mod.rs:21-53): in_off[sid] .. in_off[sid + 1] is what is resident NOW; when the host has moved the window up by
// `in_slide` bytes (a multiple of 16) since the last slice, the parked cursor moves down with it.  A slice pauses early when its
// cursor comes within a margin of the resident end while the source has more (`in_low`); a segment that still runs into the end -- a
// single command or header longer than the margin -- is taken back by the kernel itself (BRX_ST_RESTORE in brx_kernels.hip): the
// slice pauses in FRONT of it and the next one runs it with more input resident.  The record is written at a pause only; a slice
// that ends for good marks it finished (state 2) and leaves `lds` alone.  An item that does not fit the output window's capacity is taken
// back the same way and the pause reports `need_room` (round 5: before, such a slice ended with "capacity too small").
struct BrxResume {
    uint32_t state;  // 0 = fresh stream, 1 = paused (lds valid), 2 = finished
    uint32_t phase;  // where to resume (kernel-internal)
    uint64_t pause_at;
    uint64_t in_slide;
    uint64_t in_low; // pause as well once the input cursor (bits from the window's first dword) is at or beyond this: the source has
                     // more, and what is resident ends soon (~0 = the resident input is all there is)
    uint64_t need_room; // written at a pause: 0, or the output position the NEXT item (a command, an uncompressed meta-block) runs to --
                        // it did not fit the window's capacity and was taken back; the host makes that room and goes on
                        // (bit 63: the item is a whole compressed meta-block the slice paused in FRONT of -- room for it lets the assembly loop
                        // run it; without, it decodes command by command.  On the way IN: 1 = do not pause in front of meta-blocks)
    uint32_t lds[2560];
};
#define BRX_RESUME_CURSOR_WORD (2432u + 3u) // index into BrxResume::lds of the parked input cursor (Lds::st[3..4], bits from the
                                            // window's first dword): 2048 B ring + 6912 B tables + 768 B scratch, st[] follows
#define BRX_PAUSED 28 // per-stream status of a paused resumable decode (never leaves brx_api.cpp)

struct BrxKernelArgs {
    const uint8_t *in;
    const uint64_t *in_off;
    uint8_t *out;
    const uint64_t *out_off;
    uint64_t *out_len;
    int32_t *status;
    uint32_t n;
    const uint32_t *order;  // work-queue order (queue slot -> stream index), nullptr = identity
    uint32_t debug_stop;    // 0 = normal; >0 = bring-up bisection points in the kernel
    uint32_t *work_counter; // this launch's own 128-B line (ring in brx_ctx), words 0..31 zeroed in-stream before the launch:
                            // [k] ticket counter of the level-k kernel (k = 0..3), [4] tickets of the catch-all launch, [5 + j]
                            // streams in list j (j = 0..2: classified for level j + 1; j = 3: the late list), [9] tickets of the
                            // classification pre-pass, [10] streams the lean kernel listed, [11] bytes decoded twice (see below),
                            // [12] / [13] pre-pass: sizes (units of 64 B) / number of the streams that stay with the regular kernel,
                            // [14] plan B: workgroups of the wider kernels that have started, [15] lean kernel: sizes of the streams
                            // it listed (units of 64 B), [16] / [17] the largest of those sizes / 0xfffff - the smallest (lean
                            // kernel and pre-pass: which walks over an oversubscribed queue can have members at all), [18] meta-blocks
                            // taken back after a speculative end, [19] entries of the second late list (level 3 -> 4), [20] tickets
                            // of the level-4 launch
    // Streams whose meta-block tables spill a kernel's LDS table memory are not decoded there: they are LISTED for the level
    // whose table memory holds them (the need is known exactly once the header is parsed) and decoded by that level's kernel.
    //   lists 0..2 (region j of `defer`): streams of level j + 1 that have produced no output yet -- decoded from their start;
    //   list 3, the LATE list: streams that were under way when a LATER meta-block outgrew their level -- entry k comes with
    //            a state record (`handup`, 16 words per entry: the reference's Decompressor carries exactly this across a
    //            meta-block boundary, src/lib.rs:1572-1573) and is resumed at that meta-block's header, never restarted.
    //            Word 11 of the counter line adds the output position at every hand-up and subtracts it at every resume with
    //            state: 0 after the launch = no byte was decoded twice.
    // Two launch plans (launch() in brx_api.cpp): A -- the regular kernel classifies while it decodes, ONE catch-all level-3
    // launch behind it takes all four lists; B -- a header-only pre-pass of the regular kernel (`prepass`) classifies every
    // stream first (`cls`), then all four levels run NEXT TO each other on four HIP streams, each on its own list, and the
    // catch-all takes the late list.  No kernel ever waits for another one's list.
    uint32_t *defer;        // nullptr (no hand-up at all: spilled tables stay in a slab), or BRX_LIST_REGIONS regions of defer_cap words
    uint32_t defer_cap;
    uint32_t *handup;       // state records of the late list, 16 words per entry, late_cap entries
    uint32_t late_cap;
    uint32_t *handup2;      // the SECOND late list, level 3 -> level 4: state records only (the stream index is a record's word 12),
    uint32_t late2_cap;     // count in word 19 of the counter line; word 20 = tickets of the level-4 launch.  nullptr / 0: level 3 keeps
                            // a meta-block whose tables spill even there (slab + C++ loop, as before round 5)
    uint8_t *cls;           // plan B: per-stream class written by the pre-pass (0 = the regular kernel's), else nullptr
    uint32_t prepass;       // regular kernel only: 1 = classify (first meta-block header) and write cls / lists 0..2, decode nothing
    uint32_t list_mask;     // wider kernels: the lists this launch decodes (bit j = list j)
    uint32_t counter_idx;   // the word of the counter line this launch takes its tickets from
    uint32_t big_bytes;     // regular kernel, plan B: 0, or the mean compressed size of its streams -- an oversubscribed queue is then walked
                            // four times: sizes >= 2 x this, >= this, >= half of it, the rest (the long jobs start first, no sort);
                            // the pre-pass leaves the sum (in units of 64 B) and the count in words 12 / 13 of the counter line
    uint32_t start_total;   // plan B, wider kernels: the workgroup that brings word 14 of the counter line to this value writes
    uint32_t start_value;   // start_value to *start_flag (pinned host memory): every workgroup launched so far is resident, the host
    volatile uint32_t *start_flag; // may launch the next, narrower kernel (nullptr: no such hand-shake)
    uint32_t late_only;     // kernels below level 3: 1 = every hand-up goes to the late list (plan B: the class lists are being read)
    uint32_t tiny_bytes;    // compressed streams up to this size run their commands in the C++ loop alone (BRX_TINY_STREAM_BYTES)
    uint32_t sw_threshold;  // wider kernels: up to this many listed streams they run the sparse-launch build of the loop
    const BrxSlabPool *pool; // spill slabs
    unsigned long long *trace; // nullptr, or 4 words per stream (BRX_OPTION_TRACE): when and where it was decoded -- start and end
                            // (100 MHz realtime counter), HW_ID | level << 32, spare
    unsigned long long *debug; // bring-up profiling (BRX_DEBUG_STATS=1): 10 words per stream, else nullptr
    uint32_t *dump;         // bring-up (BRX_DEBUG_DUMP, debug_stop 9): word 0 = records written, then records of
                            // BRX_DUMP_WORDS words: {stream id, command index, 14 spare, the wave's whole LDS}
    uint32_t dump_interval, dump_max;
    BrxResume *resume;      // nullptr, or one record per stream: resumable mode (see BrxResume)
    uint8_t *out_mirror;    // nullptr, or the device-visible address of pinned host memory laid out like `out`: output bytes are
                            // stored to both (the D2H copy fused into the decode)
    uint32_t launch_seq;    // sequence number of this launch on its context, and
    volatile uint32_t *handed_seq; // nullptr, or a pinned host word that takes launch_seq whenever a stream is listed for a wider
                            // level: the host picks plan B only for contexts that met such streams lately
    uint32_t loop_build;    // which build of the assembly loop: 0 = bit window in VGPRs (full CUs), 1 = in SGPRs (sparse launch)
    // The lean instance in front of the regular kernel (brx_launch_decode_s): it decodes the streams of at most `small_bytes`
    // compressed bytes and lists the others (and the small ones it gives up on) in `s_list`, count in word 10 of the counter
    // line; the regular kernel then takes queue slots [0, n) from `order` / the identity as before and slots beyond from that
    // list.  Device-pointer path: n = 0 for the regular kernel, the lean one classifies all streams (`classify`: the listing of
    // the large ones is done 64 streams per atomic by its first waves).  s_list == nullptr: no lean kernel in this launch.
    uint32_t *s_list;
    uint32_t small_bytes;
    uint32_t classify;
    uint32_t n_total;       // streams of the batch (bound of s_list)
    BrxDeviceTables t;
};

#ifdef BRX_SMALL
#define BRX_LDS_BYTES 5120u
#else
#define BRX_LDS_BYTES (10240u + BRX_LDS_GROW)
#endif
#define BRX_DUMP_WORDS (16u + BRX_LDS_BYTES / 4u)

void brx_launch_decode(const BrxKernelArgs &args, unsigned grid, void *hip_stream);
void brx_launch_decode_l1(const BrxKernelArgs &args, unsigned grid, void *hip_stream); // the wider instances (args.defer set)
void brx_launch_decode_l2(const BrxKernelArgs &args, unsigned grid, void *hip_stream);
void brx_launch_decode_l3(const BrxKernelArgs &args, unsigned grid, void *hip_stream);
void brx_launch_decode_l4(const BrxKernelArgs &args, unsigned grid, void *hip_stream); // (reads args.handup2 only)
void brx_launch_decode_s(const BrxKernelArgs &args, unsigned grid, void *hip_stream);  // the lean instance (args.s_list set)
