// brx_api.cpp -- host side of the C ABI declared in include/brx.h.
//
// Owns the per-GPU context (device copies of the constant tables, the spill-slab pool, HIP streams and
// events, pinned/device staging) and implements brx_decode_batch + the Read-shaped stream facade on top of
// the gfx950 kernel in brx_kernels.hip.  There is deliberately NO CPU decode path in this library: without
// a HIP device brx_ctx_create fails with BRX_ERR_NO_DEVICE.
//
// Concurrency contract (also stated in brx.h): every entry point may be called from any thread; calls on
// one context are serialised on the host by a mutex, launches on one context never share a work counter
// (ring of counters) or a spill slab (slabs are claimed by the waves themselves from a pool), so two
// BRX_MEM_DEVICE calls on two caller HIP streams may overlap on the device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/brx.h"
#include "_gen/brx_tables_gen.h" // BRX_DICT, BRX_CONTEXT_LUT, BRX_TRANSFORMS  (tools/bin2h.py from tables/*.bin)
#include "brx_device.h"
#include "brx_internal.h"
#include "brx_plan.h"

static thread_local std::string g_err;

static int fail(int code, const char *what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(buf, sizeof buf, "%s", what);
    try {
        g_err = buf;
    } catch (...) {
    }
    return code;
}

#define HIP_TRY(call)                                              \
    do {                                                           \
        hipError_t e_ = (call);                                    \
        if (e_ != hipSuccess) return fail(BRX_ERR_HIP, #call, e_); \
    } while (0)

// Every extern "C" body runs inside this: no C++ exception may cross the C ABI.
#define BRX_GUARD_BEGIN try {
#define BRX_GUARD_END(ret_oom, ret_other)                                       \
    }                                                                           \
    catch (const std::bad_alloc &) {                                            \
        fail(BRX_ERR_OUT_OF_MEMORY, "host allocation failed");                  \
        return ret_oom;                                                         \
    }                                                                           \
    catch (...) {                                                               \
        fail(BRX_ERR_HIP, "unexpected C++ exception inside libbrx");            \
        return ret_other;                                                       \
    }

#define BRX_COUNTER_RING 64u   // launches in flight on one context before a work counter is reused
#define BRX_MAX_CHUNKS 8u      // host-pointer pipeline: H2D / decode / D2H chunks in flight
#define BRX_STREAM_LIMIT 0xffffff00ull // per-stream output limit of the 32-bit position arithmetic

struct brx_stream;

struct brx_ctx {
    int device = 0;
    std::mutex mu;
    hipStream_t stream = nullptr;                 // decode stream of the host-pointer path / default device-path stream
    hipStream_t s_chunk[8] = {};                  // host-pointer pipeline: one stream per chunk (copy in, decode, copy out)
    uint8_t *d_dict = nullptr;
    uint8_t *d_lut = nullptr;
    BrxTransform *d_xforms = nullptr;
    uint32_t *d_iac = nullptr;
    uint32_t *d_counters = nullptr; // BRX_COUNTER_RING x 128 B
    uint64_t launch_seq = 0;
    // streams whose tables spill the regular kernel's LDS are listed here by it and decoded by the wide kernel launched
    // right behind (BrxKernelArgs::defer): BRX_COUNTER_RING lists of defer_cap stream indices, one per launch in flight
    uint32_t *d_defer = nullptr;
    size_t defer_cap = 0;
    // BRX_OPT_ORDER on the device path: the queue order of a launch lives in that launch's own slot of a ring (like its work
    // counter and its defer lists), so overlapping calls on different HIP streams never share one
    uint32_t *d_order = nullptr;
    size_t order_cap = 0;
    const uint32_t *last_counter = nullptr; // counter line of the most recent launch (brx_last_timing(ctx, 2))
    uint32_t tiny_bytes = BRX_TINY_STREAM_BYTES; // bring-up / A-B: BRX_TINY_BYTES
    bool no_defer = false; // bring-up / A-B (BRX_NO_DEFER=1): spilled meta-blocks stay in the regular kernel's C++ loop
    uint32_t small_bytes = BRX_SMALL_STREAM_BYTES; // streams up to this size go to the lean instance first (0: there is none)
    uint32_t small_waves_per_cu = 32;              // grid of the lean instance (A/B)
    // spill-slab pool: slabs are claimed by waves (atomic bitmap), sized lazily by the largest grid seen
    BrxSlabPool *d_pool = nullptr; // device copy of `pool`
    BrxSlabPool pool = {nullptr, nullptr, 0, nullptr, nullptr};
    uint64_t slab_waits = 0; // of pools this context had before (brx_last_timing 12 adds the current pool's word)
    unsigned max_grid = 0;
    unsigned grid_cap = 0;
    bool force_plan_b = false;                    // BRX_OPTION_LEVELS 2: plan B (classification pre-pass, all levels next to each other) on every launch (A/B)
    size_t reader_window = (8u << 20);            // BRX_OPTION_READER_WINDOW: compressed bytes a bounded / pulled stream keeps resident
    uint64_t stream_regrown = 0;                  // bounded streams: pauses in front of ONE item that did not fit the room behind the output window (brx_last_timing 9)
    uint64_t stream_short_slices = 0;             // slices of bounded streams that paused in front of an item the resident input did not hold (brx_last_timing 8)
    bool trace_on = false;                        // BRX_OPTION_TRACE: per-stream start / end / place of the most recent launch (brx_last_trace)
    unsigned long long *d_trace = nullptr;
    size_t trace_cap = 0, trace_n = 0;
    bool no_plan_b = false;                       // BRX_OPTION_LEVELS 0: always plan A (the catch-all level-3 launch behind the regular kernel) (A/B)
    uint32_t *h_handed = nullptr, *d_handed = nullptr; // pinned host word (and its device address): BrxKernelArgs::handed_seq
    hipStream_t s_wide[3] = {};                   // plan B: the wider kernels' own streams (levels 1..3)
    hipEvent_t ev_fork[BRX_COUNTER_RING] = {}, ev_join[BRX_COUNTER_RING][3] = {};
    uint32_t *d_handup = nullptr;                 // state records of the late lists: BRX_COUNTER_RING x BRX_LATE_CAP x 16 words
    bool level4 = true;                           // BRX_OPTION_LEVEL4: the level-4 launch behind every batch launch
    bool reader_mb_room = true;                   // BRX_OPTION_READER_MB_ROOM: bounded readers make room for whole meta-blocks (A/B)
    uint32_t *d_handup2 = nullptr;                // ... of the second late lists (level 3 -> level 4): BRX_COUNTER_RING x BRX_LATE2_CAP x 16 words
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_last = nullptr; // recorded after the most recent launch (pool growth waits for it)
    // Slab accounting (round 6): the pool always has at least as many slabs as the launches IN FLIGHT on this context have waves
    // that can claim one, so a wave never waits for a slab (brx.h, BRX_INTERNAL_WATCHDOG).  ev_done[k] is recorded behind the launch
    // that used ring slot k, inflight_waves[k] is that launch's waves until the event has been seen complete.
    hipEvent_t ev_done[BRX_COUNTER_RING] = {};
    unsigned inflight_waves[BRX_COUNTER_RING] = {};
    hipEvent_t ev_in[BRX_MAX_CHUNKS] = {}, ev_k[BRX_MAX_CHUNKS] = {};
    bool have_timing = false;
    bool any_launch = false;
    // options read once from the environment (bring-up switches)
    uint32_t debug_stop = 0;
    bool debug_stats = false, debug_stats_all = false, no_order = false, no_mirror = false;
    int loop_build = -1; // -1 = by occupancy (launch()); bring-up: BRX_LOOP_BUILD forces 0 / 1
    uint32_t dump_interval = 0, dump_max = 0; // BRX_DEBUG_DUMP=interval:max:path (with BRX_DEBUG_STOP=9)
    std::string dump_path;
    // host-mode device staging (grown on demand)
    uint8_t *st_in = nullptr, *st_out = nullptr;
    uint8_t *d_gen_header = nullptr, *st_gen = nullptr; // stream generator: the constant meta-block bits, hash tables + staging
    size_t st_gen_cap = 0;
    uint64_t *st_meta = nullptr;
    size_t st_in_cap = 0, st_out_cap = 0, st_meta_cap = 0;
    // Read facade: streams created but not yet decoded (decoded together by the first read of any of them).  Round 6: `pending` and
    // `live` have a lock of their own (qmu; order: mu before qmu), so that a stream can be made while a batch is running -- the streams
    // that host threads make in the meantime queue up and go out together as the next batch.  One reader at a time leads a batch
    // (facade_busy); the others wait on qcv for THEIR stream, not for the context.
    std::mutex qmu;
    std::condition_variable qcv;
    bool facade_busy = false;
    std::condition_variable new_cv; // a stream was queued (the leading reader's short wait for the others, below)
    size_t facade_prev_n = 0;       // streams of the batch before (under qmu)
    std::vector<brx_stream *> waitq; // readers waiting for their stream, in order of arrival (each on its own condition variable: a batch
                                     // that ends wakes the owners of ITS streams and one reader to lead the next, not everybody)
    std::vector<brx_stream *> pending;
    // every live stream object of this context, bounded ones included: brx_ctx_destroy detaches them all
    std::vector<brx_stream *> live;
    uint8_t *fa_in = nullptr, *fa_out = nullptr; // the Read facade's batches: pinned, mapped staging the kernel reads and writes in place
    size_t fa_in_cap = 0, fa_out_cap = 0;        // (brx_host_alloc; kept between batches: no page faults, no copy of unused capacity)
    uint32_t fa_readers = 0;                     // owners that still have to copy their stream out of fa_out (under qmu): the next batch
    std::condition_variable fa_cv;               // waits for them before it touches the arena
    uint64_t facade_batches = 0, facade_streams = 0; // batches decode_pending_locked launched / streams in them (brx_last_timing 14 / 15)
};

extern "C" const char *brx_last_error(void) { return g_err.c_str(); }

// (brx_internal.h: for brx_node.cpp)
int brx_fail(int code, const char *what) { return fail(code, what); }
int brx_ctx_device(const brx_ctx *c) { return c->device; }
unsigned brx_ctx_max_grid(const brx_ctx *c) { return c->max_grid; }

extern "C" const char *brx_status_str(int32_t s) {
    // 1..24: description strings of the reference, src/lib.rs:331-354 (typos are the reference's)
    static const char *const STR[28] = {
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
    return (s >= 0 && s <= 27) ? STR[s] : "unknown status";
}

static void ctx_release(brx_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &q : c->s_chunk)
        if (q) (void)hipStreamSynchronize(q);
    for (auto &q : c->s_wide)
        if (q) (void)hipStreamSynchronize(q);
    if (c->ev_last && c->any_launch) (void)hipEventSynchronize(c->ev_last);
    (void)hipFree(c->d_dict);
    (void)hipFree(c->d_lut);
    (void)hipFree(c->d_xforms);
    (void)hipFree(c->d_iac);
    (void)hipFree(c->d_counters);
    if (c->h_handed) (void)hipHostFree(c->h_handed);
    (void)hipFree(c->d_defer);
    (void)hipFree(c->d_handup);
    (void)hipFree(c->d_handup2);
    (void)hipFree(c->d_trace);
    (void)hipFree(c->d_order);
    (void)hipFree(c->d_pool);
    (void)hipFree(c->pool.bitmap);
    (void)hipFree(c->pool.slabs);
    (void)hipFree(c->st_in);
    if (c->fa_in) (void)hipHostFree(c->fa_in);
    if (c->fa_out) (void)hipHostFree(c->fa_out);
    (void)hipFree(c->st_out);
    (void)hipFree(c->st_meta);
    (void)hipFree(c->d_gen_header);
    (void)hipFree(c->st_gen);
    for (auto &ev : c->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : c->ev_in)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : c->ev_k)
        if (ev) (void)hipEventDestroy(ev);
    if (c->ev_last) (void)hipEventDestroy(c->ev_last);
    for (auto &ev : c->ev_done)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : c->ev_fork)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &evs : c->ev_join)
        for (auto &ev : evs)
            if (ev) (void)hipEventDestroy(ev);
    for (auto &q : c->s_wide)
        if (q) (void)hipStreamDestroy(q);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (auto &q : c->s_chunk)
        if (q) (void)hipStreamDestroy(q);
    delete c;
}

static int ctx_init(brx_ctx *c, int device) {
    c->device = device;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    // 16 single-wave workgroups per CU: 4 per SIMD, bounded by the ~10 KiB of LDS each one declares.
    c->max_grid = (unsigned)prop.multiProcessorCount * 16u;
#ifdef BRX_BRINGUP // (build.py with BRX_BRINGUP=1 only: statistics and LDS dumps for the emulator; the shipped library reads no environment)
    {
        const char *e;
        c->debug_stats = getenv("BRX_DEBUG_STATS") != nullptr;
        c->debug_stats_all = getenv("BRX_DEBUG_STATS_ALL") != nullptr;
        if ((e = getenv("BRX_DEBUG_STOP")) != nullptr) c->debug_stop = (uint32_t)atoi(e);
        if ((e = getenv("BRX_SMALL_BYTES")) != nullptr) c->small_bytes = std::min<uint32_t>((uint32_t)atoi(e), BRX_SMALL_MAX_BYTES);
        if ((e = getenv("BRX_DEBUG_DUMP")) != nullptr) {
            unsigned iv = 0, mx = 0;
            char path[400];
            if (sscanf(e, "%u:%u:%399s", &iv, &mx, path) == 3 && iv && mx) {
                c->dump_interval = iv;
                c->dump_max = mx;
                c->dump_path = path;
            }
        }
    }
#endif
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto &q : c->s_chunk) HIP_TRY(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    for (auto &q : c->s_wide) HIP_TRY(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(&c->d_dict, sizeof BRX_DICT));
    HIP_TRY(hipMalloc(&c->d_lut, sizeof BRX_CONTEXT_LUT));
    HIP_TRY(hipMalloc(&c->d_xforms, 121 * sizeof(BrxTransform)));
    HIP_TRY(hipMalloc(&c->d_counters, BRX_COUNTER_RING * 128u));
    HIP_TRY(hipHostMalloc((void **)&c->h_handed, 16 + 64 * BRX_COUNTER_RING, hipHostMallocMapped)); // (word 0: handed_seq; from word 4: 16 words per launch slot, plan B's counts)
    *c->h_handed = 0u;
    HIP_TRY(hipHostGetDevicePointer((void **)&c->d_handed, c->h_handed, 0));
    HIP_TRY(hipMemset(c->d_counters, 0, BRX_COUNTER_RING * 128u));
    HIP_TRY(hipMalloc(&c->d_pool, sizeof(BrxSlabPool)));
    HIP_TRY(hipMemset(c->d_pool, 0, sizeof(BrxSlabPool)));
    HIP_TRY(hipMemcpy(c->d_dict, BRX_DICT, sizeof BRX_DICT, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_lut, BRX_CONTEXT_LUT, sizeof BRX_CONTEXT_LUT, hipMemcpyHostToDevice));
    // 121 transforms, serialized in the spec as prefix\0 op suffix\0 (Appendix B)
    std::vector<BrxTransform> xf(121);
    const unsigned char *p = BRX_TRANSFORMS;
    for (int i = 0; i < 121; i++) {
        memset(&xf[i], 0, sizeof(BrxTransform));
        size_t pl = strlen((const char *)p);
        memcpy(xf[i].prefix, p, pl);
        xf[i].plen = (uint8_t)pl;
        p += pl + 1;
        xf[i].op = *p++;
        size_t sl = strlen((const char *)p);
        memcpy(xf[i].suffix, p, sl);
        xf[i].slen = (uint8_t)sl;
        p += sl + 1;
    }
    HIP_TRY(hipMemcpy(c->d_xforms, xf.data(), 121 * sizeof(BrxTransform), hipMemcpyHostToDevice));
    {
        // Insert&copy alphabet (spec section 5, reference src/lib.rs:962-976 + src/lookuptable/mod.rs:59-123): symbol ->
        // insert code / copy code through the 11 x 64 cell layout, code -> (base, extra bits).
        static const uint16_t ins_base[24] = {0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26, 34, 50, 66, 98, 130, 194, 322, 578, 1090, 2114, 6210, 22594};
        static const uint8_t ins_extra[24] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24};
        static const uint16_t cpy_base[24] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 18, 22, 30, 38, 54, 70, 102, 134, 198, 326, 582, 1094, 2118};
        static const uint8_t cpy_extra[24] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24};
        static const uint8_t cell_ins[11] = {0, 0, 0, 0, 8, 8, 0, 16, 8, 16, 16}, cell_cpy[11] = {0, 8, 0, 8, 0, 8, 16, 0, 16, 8, 16};
        static const uint8_t ndbits[25] = {0, 0, 0, 0, 10, 10, 11, 11, 10, 10, 10, 10, 10, 9, 9, 8, 7, 7, 8, 7, 7, 6, 6, 5, 5};
        // record of symbol s = dwords 4s .. 4s+3 (one s_load_dwordx4 of the assembly loop, straight into its INS / CPY /
        // DCTX registers): insert base, copy base, 2 * distance context (min(copy_len - 2, 3), a function of the copy
        // code; 4 for an implicit distance code 0, src/lib.rs:2012-2015 -- doubled: the loop uses it as the lane of the tree's
        // descriptor and as the offset of the tree's register pair), insert extra bits | copy extra bits << 8
        std::vector<uint32_t> t(704 * 4 + 64, 0u);
        for (unsigned sym = 0; sym < 704; sym++) {
            unsigned cell = sym >> 6, ic = cell_ins[cell] + ((sym >> 3) & 7u), cc = cell_cpy[cell] + (sym & 7u);
            t[4 * sym] = ins_base[ic];
            t[4 * sym + 1] = cpy_base[cc];
            t[4 * sym + 2] = 2u * (sym < 128 ? 4u : (cc < 3u ? cc : 3u));
            t[4 * sym + 3] = ins_extra[ic] | ((uint32_t)cpy_extra[cc] << 8);
        }
        uint32_t off = 0;
        for (unsigned n = 0; n < 25; n++) { // DOFFSET: words of length n start here (spec section 8)
            t[2816 + n] = off | ((uint32_t)ndbits[n] << 24);
            if (n >= 4) off += n << ndbits[n];
        }
        if (off != sizeof BRX_DICT) return fail(BRX_ERR_HIP, "internal: dictionary size does not match NDBITS");
        HIP_TRY(hipMalloc(&c->d_iac, t.size() * 4));
        HIP_TRY(hipMemcpy(c->d_iac, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    }
    for (auto &ev : c->ev) HIP_TRY(hipEventCreate(&ev));
    for (auto &ev : c->ev_in) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (auto &ev : c->ev_k) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_last, hipEventDisableTiming));
    for (auto &ev : c->ev_done) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (auto &ev : c->ev_fork) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (auto &evs : c->ev_join)
        for (auto &ev : evs) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    return BRX_SUCCESS;
}

extern "C" int brx_ctx_create(brx_ctx **out, int device) {
    BRX_GUARD_BEGIN
    if (!out) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(BRX_ERR_NO_DEVICE, "no HIP device: libbrx has no CPU fallback", e);
    if (device < 0 || device >= ndev) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_ctx_create: bad device index");
    HIP_TRY(hipSetDevice(device));
    brx_ctx *c = new (std::nothrow) brx_ctx();
    if (!c) return fail(BRX_ERR_OUT_OF_MEMORY, "brx_ctx_create: host allocation failed");
    int rc = ctx_init(c, device);
    if (rc != BRX_SUCCESS) {
        std::string keep = g_err;
        ctx_release(c); // nothing leaks on a failed create
        g_err = keep;
        return rc;
    }
    *out = c;
    return BRX_SUCCESS;
    BRX_GUARD_END(BRX_ERR_OUT_OF_MEMORY, BRX_ERR_HIP)
}

static void stream_detach(brx_stream *s);

extern "C" int brx_ctx_set_option(brx_ctx *c, uint32_t option, int64_t value) {
    if (!c) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_ctx_set_option: ctx is NULL");
    std::lock_guard<std::mutex> lk(c->mu);
    switch (option) {
    case BRX_OPTION_COMMAND_LOOP:
        if (value != 0 && value != 6 && value != 7 && value != 8) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_ctx_set_option: command loop is 0, 6, 7 or 8");
        c->debug_stop = (uint32_t)value;
        break;
    case BRX_OPTION_LOOP_BUILD:
        if (value < -1 || value > 1) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_ctx_set_option: loop build is -1, 0 or 1");
        c->loop_build = (int)value;
        break;
    case BRX_OPTION_QUEUE_ORDER: c->no_order = value == 0; break;
    case BRX_OPTION_HAND_UP: c->no_defer = value == 0; break;
    case BRX_OPTION_LEVELS:
        if (value < 0 || value > 2) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_ctx_set_option: levels is 0, 1 or 2");
        c->no_plan_b = value == 0;
        c->force_plan_b = value >= 2;
        break;
    case BRX_OPTION_TINY_BYTES: c->tiny_bytes = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 1 << 20); break;
    case BRX_OPTION_HOST_IN_PLACE: c->no_mirror = value == 0; break;
    case BRX_OPTION_GRID_CAP: c->grid_cap = (unsigned)std::max<int64_t>(value, 0); break;
    case BRX_OPTION_SMALL_BYTES: c->small_bytes = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), BRX_SMALL_MAX_BYTES); break;
    case BRX_OPTION_TRACE: c->trace_on = value != 0; break;
    case BRX_OPTION_LEVEL4: c->level4 = value != 0; break;
    case BRX_OPTION_READER_MB_ROOM: c->reader_mb_room = value != 0; break;
    case BRX_OPTION_READER_WINDOW:
        if (value < (1 << 20) || value > (256 << 20)) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_ctx_set_option: reader window is 1 MiB .. 256 MiB");
        c->reader_window = ((size_t)value + 65535u) & ~(size_t)65535;
        break;
    case BRX_OPTION_SMALL_WAVES: c->small_waves_per_cu = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 1), 32); break;
    default: return fail(BRX_ERR_INVALID_ARGUMENT, "brx_ctx_set_option: unknown option");
    }
    return BRX_SUCCESS;
}

extern "C" void brx_ctx_destroy(brx_ctx *c) {
    if (!c) return;
    try {
        std::vector<brx_stream *> orphans;
        {
            std::lock_guard<std::mutex> lk(c->mu);
            std::lock_guard<std::mutex> ql(c->qmu);
            orphans.swap(c->live);
            c->pending.clear();
        }
        for (brx_stream *s : orphans) stream_detach(s); // (bounded streams too: their device buffers go with the context)
    } catch (...) {
    }
    ctx_release(c);
}

static int grow(uint8_t **p, size_t *cap, size_t need) {
    if (need <= *cap && *p) return BRX_SUCCESS;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    size_t want = need + need / 4 + 4096;
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) return fail(BRX_ERR_OUT_OF_MEMORY, "device staging allocation failed", e);
    *cap = want;
    return BRX_SUCCESS;
}

// The spill-slab pool serves tables that do not fit the LDS table memory (worst case 256 trees per category,
// SURVEY 2.2): 896 KiB per slab, claimed by a wave the first time it spills and released at the end of its
// stream.  One slab per wave of the launches in flight (pool_need) means no wave ever waits for one;
// nothing is allocated before the first launch and small batches stay small.
static int ensure_pool(brx_ctx *c, unsigned grid) {
    unsigned want = 64;
    while (want < grid) want <<= 1;
    if (want > c->max_grid) want = ((c->max_grid + 31u) / 32u) * 32u + 32u; // (what the chip holds at a time, and a word to spare)
    if (c->pool.slabs && c->pool.count >= want) return BRX_SUCCESS;
    HIP_TRY(hipDeviceSynchronize()); // (rare) nobody may hold a slab of the old pool: launches on any stream included
    for (auto &w : c->inflight_waves) w = 0u; // (nothing is in flight any more)
    if (c->pool.waits) {
        uint32_t w = 0;
        if (hipMemcpy(&w, c->pool.waits, 4, hipMemcpyDeviceToHost) == hipSuccess) c->slab_waits += w;
    }
    (void)hipFree(c->pool.bitmap);
    (void)hipFree(c->pool.slabs);
    c->pool.bitmap = nullptr;
    c->pool.slabs = nullptr;
    c->pool.waits = nullptr;
    c->pool.count = 0;
    hipError_t e = hipMalloc(&c->pool.slabs, ((size_t)want + 1u) * BRX_SCRATCH_WORDS * 4u); // (+ 1: BrxSlabPool::sink)
    if (e != hipSuccess) return fail(BRX_ERR_OUT_OF_MEMORY, "spill-slab pool allocation failed", e);
    e = hipMalloc(&c->pool.bitmap, (size_t)(want / 32u + 1u) * 4u); // (+ 1: BrxSlabPool::waits)
    if (e != hipSuccess) return fail(BRX_ERR_OUT_OF_MEMORY, "spill-slab bitmap allocation failed", e);
    HIP_TRY(hipMemset(c->pool.bitmap, 0, (size_t)(want / 32u + 1u) * 4u));
    c->pool.count = want;
    c->pool.waits = c->pool.bitmap + want / 32u;
    c->pool.sink = c->pool.slabs + (size_t)want * BRX_SCRATCH_WORDS;
    HIP_TRY(hipMemcpy(c->d_pool, &c->pool, sizeof(BrxSlabPool), hipMemcpyHostToDevice));
    return BRX_SUCCESS;
}

// How many slabs the pool must have for a launch whose kernels run `waves` slab-claiming waves at a time: those, plus the waves
// of every earlier launch of this context that may still be running (device-pointer calls on other HIP streams overlap).  A slab is
// held by a RESIDENT wave only -- claimed at a header, released before the wave takes its next stream -- and the chip holds at most
// max_grid of them (16 per CU: 10 KiB of LDS, 128 VGPRs), whatever is queued: that many slabs can never run out.  Until round 6 the
// pool followed the largest single grid, a wave of a second overlapping launch could find every slab taken, and after 4 s it gave up:
// a VALID stream came back with BRX_INTERNAL_WATCHDOG.  Events are only polled when the bound without polling
// (every launch of the last 64 counted as running) exceeds the pool: synchronous callers never poll, back-to-back asynchronous ones
// about once per launch.
static unsigned pool_need(brx_ctx *c, unsigned waves) {
    const unsigned cap = ((c->max_grid + 31u) / 32u) * 32u + 32u;
    if (c->pool.slabs && c->pool.count >= cap) return cap;
    unsigned busy = 0;
    for (auto w : c->inflight_waves) busy += w;
    if (busy + waves > c->pool.count && busy != 0u) {
        busy = 0;
        for (unsigned k = 0; k < BRX_COUNTER_RING; k++) {
            if (c->inflight_waves[k] == 0u) continue;
            if (hipEventQuery(c->ev_done[k]) == hipSuccess) c->inflight_waves[k] = 0u;
            else busy += c->inflight_waves[k];
        }
        (void)hipGetLastError(); // (hipErrorNotReady is an answer, not a failure of this call)
    }
    return std::min(cap, busy + waves + 32u); // (+ a word to spare: the last claimer does not have to hunt for the one free slab)
}

// Streams per CU up to which a launch counts as sparse (profiles/r02_loop_build_sweep.txt).
#define BRX_SW_WAVES_PER_CU 6u
// Batches beyond this many streams keep their spilling streams in the regular kernel (the lists would be 64 x 4 B x n).
#define BRX_DEFER_MAX_STREAMS (1u << 18)

// Room for the lists of n stream indices each of every launch in flight (grown rarely: nobody may be using the old lists).
static int ensure_defer(brx_ctx *c, uint32_t n) {
    if (c->d_defer && c->d_handup && c->d_handup2 && c->defer_cap >= n) return BRX_SUCCESS;
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(c->d_defer);
    c->d_defer = nullptr;
    c->defer_cap = 0;
    size_t cap = 4096;
    while (cap < n) cap <<= 1;
    hipError_t e;
    if (!c->d_handup) { // (first: a launch must never see lists without the state records that go with them)
        e = hipMalloc(&c->d_handup, (size_t)BRX_COUNTER_RING * BRX_LATE_CAP * 16u * 4u);
        if (e != hipSuccess) { c->d_handup = nullptr; return fail(BRX_ERR_OUT_OF_MEMORY, "state-record allocation failed", e); }
    }
    if (!c->d_handup2) {
        e = hipMalloc(&c->d_handup2, (size_t)BRX_COUNTER_RING * BRX_LATE2_CAP * 16u * 4u);
        if (e != hipSuccess) { c->d_handup2 = nullptr; return fail(BRX_ERR_OUT_OF_MEMORY, "state-record allocation failed", e); }
    }
    e = hipMalloc(&c->d_defer, cap * 4u * BRX_LIST_REGIONS * BRX_COUNTER_RING); // (brx_device.h: lists 0..2, late, lean, class bytes)
    if (e != hipSuccess) { c->d_defer = nullptr; return fail(BRX_ERR_OUT_OF_MEMORY, "deferred-stream list allocation failed", e); }
    c->defer_cap = cap;
    return BRX_SUCCESS;
}

// Room for one queue order of n stream indices per launch in flight (BRX_OPT_ORDER, device path).
static int ensure_order(brx_ctx *c, uint32_t n) {
    if (c->d_order && c->order_cap >= n) return BRX_SUCCESS;
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(c->d_order);
    c->d_order = nullptr;
    c->order_cap = 0;
    size_t cap = 4096;
    while (cap < n) cap <<= 1;
    hipError_t e = hipMalloc(&c->d_order, cap * 4u * BRX_COUNTER_RING);
    if (e != hipSuccess) return fail(BRX_ERR_OUT_OF_MEMORY, "queue-order allocation failed", e);
    c->order_cap = cap;
    return BRX_SUCCESS;
}

static int launch(brx_ctx *c, hipStream_t st, bool timing, const uint8_t *d_in, const uint64_t *d_in_off, uint32_t n,
                  uint8_t *d_out, const uint64_t *d_out_off, uint64_t *d_out_len, int32_t *d_status,
                  const uint32_t *d_order = nullptr, BrxResume *d_resume = nullptr, const BrxSlabPool *d_own_pool = nullptr,
                  uint8_t *d_out_mirror = nullptr, bool may_plan_b = false, uint32_t n_large = 0xffffffffu) {
    BrxKernelArgs a;
    a.out_mirror = d_out_mirror;
    a.order = d_order;
    a.in = d_in;
    a.in_off = d_in_off;
    a.out = d_out;
    a.out_off = d_out_off;
    a.out_len = d_out_len;
    a.status = d_status;
    a.n = n;
    a.debug_stop = c->debug_stop;
    unsigned grid = n < c->max_grid ? n : c->max_grid;
    if (c->grid_cap != 0u && grid > c->grid_cap) grid = c->grid_cap; // (A/B knob BRX_GRID_CAP: fewer resident waves, more rounds)
    a.pool = d_own_pool;
    // (what this launch runs at a time: its regular grid; under plan B the wider grids next to it -- together never more than its
    // streams, nor than the chip holds; the grid-cap knob caps each of the four grids)
    const bool can_plan_b = may_plan_b && !c->no_defer && !c->no_plan_b && d_resume == nullptr && c->debug_stop == 0u && n <= BRX_DEFER_MAX_STREAMS;
    const unsigned slab_waves = d_own_pool ? 0u : (c->grid_cap != 0u && can_plan_b) ? std::min(n, 4u * c->grid_cap) : grid;
    if (!d_own_pool) {
        int rc = ensure_pool(c, pool_need(c, slab_waves));
        if (rc) return rc;
        a.pool = c->d_pool;
    }
    a.resume = d_resume;
    // which build of the command loop (brx_hot.S): with at most BRX_SW_WAVES_PER_CU streams per CU the CU's scalar ALU has
    // room and the build with the shorter dependent chain wins; fuller CUs take the one that spares the scalar ALU
    a.loop_build = c->loop_build >= 0 ? (uint32_t)c->loop_build : (grid <= c->max_grid / 16u * BRX_SW_WAVES_PER_CU ? 1u : 0u);
    const size_t ring_slot = (size_t)(c->launch_seq++ % BRX_COUNTER_RING);
    a.work_counter = c->d_counters + ring_slot * 32u; // one 128-B line per launch
    // streams whose tables spill a kernel's LDS table memory are listed for the level that holds them (not in the resumable and
    // bring-up modes): BrxKernelArgs::defer
    a.tiny_bytes = c->tiny_bytes;
    a.defer = nullptr;
    a.defer_cap = 0;
    a.handup = nullptr;
    a.late_cap = 0;
    a.handup2 = nullptr;
    a.late2_cap = 0;
    a.cls = nullptr;
    a.prepass = 0u;
    a.list_mask = 0u;
    a.counter_idx = 0u;
    a.late_only = 0u;
    a.big_bytes = 0u;
    a.start_total = 0u;
    a.start_value = 0u;
    a.start_flag = nullptr;
    a.sw_threshold = c->loop_build >= 0 ? 0u : c->max_grid / 16u * BRX_SW_WAVES_PER_CU;
    uint32_t *regions = nullptr; // this launch's BRX_LIST_REGIONS regions of defer_cap words
    if (!c->no_defer && d_resume == nullptr && c->debug_stop == 0u && n <= BRX_DEFER_MAX_STREAMS) {
        int rc = ensure_defer(c, n);
        if (rc) return rc;
        regions = c->d_defer + ring_slot * BRX_LIST_REGIONS * c->defer_cap;
        a.defer = regions;
        a.defer_cap = (uint32_t)c->defer_cap;
        a.handup = c->d_handup + ring_slot * (size_t)BRX_LATE_CAP * 16u;
        a.late_cap = (uint32_t)std::min<size_t>(c->defer_cap, BRX_LATE_CAP);
        if (c->level4) { // (without it level 3 keeps what spills it: BrxKernelArgs::handup2)
            a.handup2 = c->d_handup2 + ring_slot * (size_t)BRX_LATE2_CAP * 16u;
            a.late2_cap = BRX_LATE2_CAP;
        }
    }
    // The lean instance in front (brx_small.h; 32 waves per CU): it decodes the streams of at most BRX_SMALL_STREAM_BYTES
    // compressed bytes and lists the rest -- and what it gives up on -- for the regular kernel (BrxKernelArgs::s_list).  With a
    // queue order from the host (longest first) the small streams are the order's tail: `n_large` says where it starts, the
    // regular kernel keeps the head as its own queue.  Without one the lean kernel classifies all n streams itself and the
    // regular kernel's whole queue is that list.
    a.s_list = nullptr;
    a.small_bytes = c->small_bytes;
    a.classify = 0u;
    a.n_total = n;
    bool lean = false, lean_tail = false;
    if (c->small_bytes != 0u && d_resume == nullptr && c->debug_stop == 0u && n <= BRX_DEFER_MAX_STREAMS && (d_order == nullptr || n_large < n || n_large == 0xffffffffu)) {
        int rc = ensure_defer(c, n);
        if (rc) return rc;
        a.s_list = c->d_defer + (ring_slot * BRX_LIST_REGIONS + 4u) * c->defer_cap;
        lean_tail = d_order != nullptr && n_large <= n; // (else: the lean kernel classifies, the regular one takes the list)
        if (lean_tail) {
            a.n = n_large;
        } else {
            a.order = nullptr;
            a.n = 0u;
        }
        lean = true;
    }
    a.debug = nullptr;
    a.trace = nullptr;
    if (c->trace_on) { // diagnostics: one record per stream
        if (c->trace_cap < n) {
            HIP_TRY(hipDeviceSynchronize());
            (void)hipFree(c->d_trace);
            c->d_trace = nullptr;
            c->trace_cap = 0;
            if (hipMalloc(&c->d_trace, (size_t)n * 32u) != hipSuccess) return fail(BRX_ERR_OUT_OF_MEMORY, "trace allocation failed");
            c->trace_cap = n;
        }
        HIP_TRY(hipMemsetAsync(c->d_trace, 0, (size_t)n * 32u, st));
        a.trace = c->d_trace;
        c->trace_n = n;
    }
    a.dump = nullptr;
    a.dump_interval = c->dump_interval;
    a.dump_max = c->dump_max;
#ifdef BRX_BRINGUP // (build.py with BRX_BRINGUP=1: per-stream statistics, LDS dumps for tools/asm_emu.py)
    unsigned long long *dbg = nullptr;
    if (c->debug_stats) {
        if (hipMalloc(&dbg, (size_t)n * (80 + 256)) == hipSuccess) { (void)hipMemset(dbg, 0, (size_t)n * (80 + 256)); a.debug = dbg; }
    }
    const size_t dump_bytes = (16u + (size_t)c->dump_max * BRX_DUMP_WORDS) * 4u;
    if (c->dump_max && c->debug_stop == 9u && hipMalloc(&a.dump, dump_bytes) == hipSuccess) (void)hipMemset(a.dump, 0, 64);
#endif
    a.t.dict = c->d_dict;
    a.t.context_lut = c->d_lut;
    a.t.xforms = c->d_xforms;
    a.t.iac = c->d_iac;
    // Two launch plans for the wider levels (12 / 8 / 4 waves per CU; brx_device.h):
    //   A  the regular kernel classifies on its way (a stream whose first header spills is listed for the level that holds its
    //      tables), ONE catch-all level-3 launch behind it takes every list.  An empty launch costs ~5 us, and most batches
    //      never list anything.
    //   B  a header-only pre-pass of the regular kernel classifies every stream of its queue first; the host reads the counts
    //      and launches exactly the kernels that have work, NEXT TO each other on their own HIP streams, each on its own complete
    //      list from its first wave on -- in a mixed batch the streams of the wider levels would otherwise wait for the longest
    //      stream of the regular kernel before they even start; the caller's stream joins them, and the catch-all takes the late
    //      list (streams that outgrew their level at a later meta-block: resumed there with their state).
    // B costs the pre-pass (the first header of every stream is parsed twice: 0.3 .. 1 ms for a full grid, and the call waits
    // for it) and the fork / join events, so it is taken only by contexts that listed a stream within their last 64 launches:
    // every kernel that lists one notes the launch in a pinned host word, read here without any API call.  No kernel ever waits
    // for another one's list, and correctness does not depend on the plan.
    a.launch_seq = (uint32_t)c->launch_seq; // (already advanced: >= 1)
    a.handed_seq = c->d_handed;
    const uint32_t seen = c->h_handed ? *(volatile uint32_t *)c->h_handed : 0u;
    const bool lately = c->force_plan_b || (seen != 0u && a.launch_seq - seen <= 64u);
    const bool plan_b = a.defer != nullptr && may_plan_b && !c->no_plan_b && lately;
    HIP_TRY(hipMemsetAsync(a.work_counter, 0, 128, st));
    if (timing) HIP_TRY(hipEventRecord(c->ev[2], st));
    if (lean) {
        BrxKernelArgs as = a;
        if (lean_tail) { // the order's tail
            as.order = d_order + n_large;
            as.n = n - n_large;
        } else {
            as.n = n;
            as.classify = 1u;
        }
        brx_launch_decode_s(as, std::min(as.n, c->max_grid / 16u * c->small_waves_per_cu), st);
    }
    const unsigned per_cu = c->max_grid / 16u;
    bool joined[3] = {false, false, false};
    if (plan_b) {
        BrxKernelArgs ap = a;
        ap.cls = (uint8_t *)(regions + 5u * c->defer_cap);
        ap.prepass = 1u;
        ap.counter_idx = 9u;
        brx_launch_decode(ap, grid, st);
        // The host reads the pre-pass's counts (lists 0..2, and how many streams the lean kernel left to the regular one) and
        // launches exactly the kernels that have work, with exactly their grids: the call waits here for the pre-pass (a header
        // per stream: 0.3 .. 1 ms for a full grid).  Workgroups that would only find their list empty are not harmless next
        // to real ones: a CU's LDS is handed out first-fit, and 10 KiB workgroups that come and go while 20 KiB ones are being
        // placed leave those at offsets between which nothing of their size fits any more (4096 x mapsdatazrh: 6 instead of 8
        // level-2 streams per CU for the whole launch, 89 ms instead of 60).
        uint32_t *hc = c->h_handed + 4u + 16u * (uint32_t)ring_slot;
        HIP_TRY(hipMemcpyAsync(hc, a.work_counter + 5, 36, hipMemcpyDeviceToHost, st)); // words 5 .. 13
        HIP_TRY(hipEventRecord(c->ev_fork[ring_slot], st));
        HIP_TRY(hipEventSynchronize(c->ev_fork[ring_slot]));
        const uint32_t cnt[4] = {0u, std::min<uint32_t>(hc[0], n), std::min<uint32_t>(hc[1], n), std::min<uint32_t>(hc[2], n)};
        const uint32_t queued = a.n + (a.s_list != nullptr ? std::min<uint32_t>(hc[5], n) : 0u); // the regular kernel's queue
        const uint32_t wide = cnt[1] + cnt[2] + cnt[3];
        const uint32_t n0 = queued > wide ? queued - wide : 0u;
        a.cls = ap.cls;
        a.late_only = 1u;
        // the regular kernel's queue, long jobs first without a sort: two walks split at the mean compressed size of its streams
        a.big_bytes = hc[8] != 0u ? (uint32_t)std::min<uint64_t>(((uint64_t)hc[7] << 6) / hc[8], 0xffffffffull) : 0u;
        // What is resident together: brx_plan.h (persistent grids that all find room at once, by 40-KiB LDS parts)
        const BrxPlanB plan = brx_plan_b(n0, cnt, c->max_grid / 4u, c->max_grid);
        uint32_t g[4] = {std::min<uint32_t>(plan.grid[0], grid), plan.grid[1], plan.grid[2], plan.grid[3]};
        if (c->grid_cap != 0u) // (the A/B knob caps EVERY grid: the slab pool follows the capped one -- a wider kernel with more waves
            for (int k = 1; k < 4; k++) g[k] = std::min<uint32_t>(g[k], c->grid_cap); // than slabs would starve its own waiters)
        const uint32_t *mask = plan.mask;
        const int narrowest = g[0] ? 0 : g[1] ? 1 : g[2] ? 2 : 3;
        uint32_t started = 0;
        hc[12] = 0u;
        for (int k = 3; k >= 1; k--) {
            if (g[k] == 0u) continue;
            BrxKernelArgs aw = a;
            aw.cls = nullptr;
            aw.list_mask = mask[k];
            aw.counter_idx = (uint32_t)k;
            started += g[k];
            aw.start_total = started;
            aw.start_value = (a.launch_seq << 2) | (uint32_t)k;
            aw.start_flag = k == narrowest ? nullptr : c->d_handed + 4u + 16u * (uint32_t)ring_slot + 12u;
            hipStream_t sw = k == narrowest ? st : c->s_wide[k - 1];
            if (sw != st) HIP_TRY(hipStreamWaitEvent(sw, c->ev_fork[ring_slot], 0));
            if (k == 1) brx_launch_decode_l1(aw, g[k], sw); else if (k == 2) brx_launch_decode_l2(aw, g[k], sw); else brx_launch_decode_l3(aw, g[k], sw);
            if (sw != st) { HIP_TRY(hipEventRecord(c->ev_join[ring_slot][k - 1], sw)); joined[k - 1] = true; }
            if (aw.start_flag != nullptr) {
                // Widest first, and the next kernel only once this one's workgroups are all resident (its last one to start says
                // so in pinned host memory; ~10 us): which of two launches on two HIP streams gets to the CUs first is a race, and
                // narrow workgroups that win it sit three or four to a 40-KiB part of the LDS until they all have left -- the
                // wider kernel then runs BEHIND them (8192 mixed streams: 60 or 110 ms from one launch to the next).  A kernel that
                // does not report within 2 ms (the chip is busy with somebody else's work) is not waited for any longer.
                const auto t0 = std::chrono::steady_clock::now();
                while (*(volatile uint32_t *)&hc[12] != aw.start_value) {
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
                }
            }
        }
        if (g[0] != 0u) brx_launch_decode(a, g[0], st);
    } else {
        brx_launch_decode(a, grid, st);
    }
    HIP_TRY(hipGetLastError());
    if (a.defer != nullptr) {
        for (int k = 0; k < 3; k++)
            if (joined[k]) HIP_TRY(hipStreamWaitEvent(st, c->ev_join[ring_slot][k], 0));
        BrxKernelArgs ac = a; // the catch-all: its waves leave at once when nothing is listed
        ac.cls = nullptr;
        ac.list_mask = plan_b ? 8u : 15u;
        ac.counter_idx = 4u;
        // (never more waves than the slab pool has slabs -- ensure_pool(grid) above: a level-3 wave whose stream spills even there
        // holds its slab for the whole decode, and a waiter gives up after 0.5 s: found by the round-5 soak with BRX_GRID_CAP = 64,
        // 199 slab-class streams on a 1024-wave catch-all over 64 slabs, tools/device_fuzz.py 3 302)
        unsigned gc = std::min(n, per_cu * 4u);
        if (c->grid_cap != 0u) gc = std::min(gc, c->grid_cap);
        brx_launch_decode_l3(ac, gc, st);
        HIP_TRY(hipGetLastError());
        // ... and behind it level 4 (150 KiB of LDS: one workgroup per CU) for what level-3 kernels handed on -- meta-blocks whose tables
        // spill even level 3 (one heterogeneous piece of > 1 MiB from the reference encoder: 10 .. 35 k words).  Its workgroups leave at once
        // when the second late list is empty.
        if (c->level4) {
            BrxKernelArgs a4 = a;
            a4.cls = nullptr;
            a4.list_mask = 8u;
            a4.counter_idx = 20u;
            unsigned g4 = std::min(n, per_cu);
            if (c->grid_cap != 0u) g4 = std::min(g4, c->grid_cap);
            brx_launch_decode_l4(a4, g4, st);
            HIP_TRY(hipGetLastError());
        }
    }
    if (timing) HIP_TRY(hipEventRecord(c->ev[3], st));
    HIP_TRY(hipEventRecord(c->ev_last, st));
    if (slab_waves != 0u) {
        HIP_TRY(hipEventRecord(c->ev_done[ring_slot], st));
        c->inflight_waves[ring_slot] = slab_waves;
    }
    c->last_counter = (a.defer != nullptr || lean) ? a.work_counter : nullptr;
    c->any_launch = true;
#ifdef BRX_BRINGUP
    if (a.dump) { // bring-up: parked decoder states for tools/asm_emu.py
        (void)hipStreamSynchronize(st);
        std::vector<uint32_t> h(dump_bytes / 4u);
        (void)hipMemcpy(h.data(), a.dump, dump_bytes, hipMemcpyDeviceToHost);
        const uint32_t nrec = std::min(h[0], c->dump_max);
        if (FILE *f = fopen(c->dump_path.c_str(), "ab")) {
            const uint64_t hdr[4] = {0x31504d5544585242ull /* "BRXDUMP1" */, nrec, (uint64_t)(uintptr_t)d_in, (uint64_t)(uintptr_t)d_out};
            fwrite(hdr, 8, 4, f);
            fwrite(h.data() + 16, 4, (size_t)nrec * BRX_DUMP_WORDS, f);
            fclose(f);
        }
        (void)hipFree(a.dump);
    }
    if (dbg) {
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h((size_t)n * 42);
        (void)hipMemcpy(h.data(), dbg, (size_t)n * (80 + 256), hipMemcpyDeviceToHost);
        static const char *nm[10] = {"hdr", "iac", "lit", "dist", "copy", "ncmd", "nlit", "fastmb", "total", "scr_top"};
        for (uint32_t i = 0; i < n && i < (c->debug_stats_all ? n : 2u); i++) {
            fprintf(stderr, "[brx stats] stream %u:", i);
            for (int q = 0; q < 10; q++) fprintf(stderr, " %s=%llu", nm[q], h[(size_t)i * 10 + q]);
            fprintf(stderr, "\n[brx stats] words:");
            for (int q = 0; q < 8; q++) fprintf(stderr, " %u %u", (unsigned)h[(size_t)i * 10 + q], (unsigned)(h[(size_t)i * 10 + q] >> 32));
            fprintf(stderr, "\n[brx phases] cycles(visits):");
            static const char *pn[16] = {"frame", "header", "gen_start", "gen_resume", "asm", "finish", "h_simple", "h_clcode", "h_clsyms",
                                         "h_build", "h_cmsyms", "h_imtf", "h_other", "dec_load", "setup", "status"};
            for (int q = 0; q < 16; q++)
                if (h[(size_t)n * 10 + (size_t)i * 32 + 16 + q])
                    fprintf(stderr, " %s=%llu(%llu)", pn[q], h[(size_t)n * 10 + (size_t)i * 32 + q], h[(size_t)n * 10 + (size_t)i * 32 + 16 + q]);
            fprintf(stderr, "\n");
        }
        (void)hipFree(dbg);
    }
#endif
    return BRX_SUCCESS;
}

// ---- host pointers: H2D, decode, D2H ----------------------------------------------------------------------
// The batch is cut into up to BRX_MAX_CHUNKS runs of consecutive streams, each with its own HIP stream: input copy,
// decode kernel, output copy.  The chunks' kernels run CONCURRENTLY on the GPU (every launch has its own work counter and
// the spill slabs are claimed per wave), so the copies of one chunk overlap the decoding of the others.  With pinned caller memory (brx_host_alloc / hipHostMalloc / hipHostRegister) the copies are true asynchronous
// DMA; with pageable memory the HIP runtime stages them and the overlap is partial.
static int decode_host(brx_ctx *c, const uint8_t *in, const uint64_t *in_off, uint32_t n, uint8_t *out,
                       const uint64_t *out_off, uint64_t *out_len, int32_t *status, bool timing) {
    for (uint32_t i = 0; i < n; i++)
        if (in_off[i + 1] < in_off[i] || out_off[i + 1] < out_off[i])
            return fail(BRX_ERR_INVALID_ARGUMENT, "brx_decode_batch: offsets must be non-decreasing");
    const uint64_t in_lo = in_off[0], in_hi = in_off[n], out_lo = out_off[0], out_hi = out_off[n];
    const size_t in_bytes = (size_t)(in_hi - in_lo), out_bytes = (size_t)(out_hi - out_lo);
    if ((in_bytes && !in) || (out_bytes && !out)) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_decode_batch: NULL data");
    int rc;
    if ((rc = grow(&c->st_out, &c->st_out_cap, out_bytes + 16))) return rc; // (always: the HBM slot is the stream's window)
    const size_t meta_words = 3 * (size_t)(n + 1);
    const size_t meta_bytes = meta_words * 8 + (size_t)n * 4 + (size_t)n * 4;
    if ((rc = grow((uint8_t **)&c->st_meta, &c->st_meta_cap, meta_bytes))) return rc;
    std::vector<uint64_t> hmeta(2 * (size_t)(n + 1));
    for (uint32_t i = 0; i <= n; i++) {
        hmeta[i] = in_off[i] - in_lo;
        hmeta[(size_t)(n + 1) + i] = out_off[i] - out_lo;
    }
    uint64_t *d_in_off = c->st_meta, *d_out_off = c->st_meta + (n + 1), *d_out_len = c->st_meta + 2 * (size_t)(n + 1);
    int32_t *d_status = (int32_t *)(c->st_meta + meta_words);
    uint32_t *d_order = (uint32_t *)(d_status + n);

    // chunk boundaries: equal shares of (input + output) bytes, and at least one full grid of streams per chunk -- a
    // stream is ~9 ms of pure latency however few run, so a batch that fits the GPU at once is fastest as ONE launch
    // (measured: 4096 x alice29, 33 ms as one chunk, 40 ms as eight); chunks pay off from the second grid-full on
    const uint64_t total = (uint64_t)in_bytes + out_bytes;
    unsigned nchunks = (unsigned)std::min<uint64_t>(BRX_MAX_CHUNKS, std::max<uint64_t>(1, n / c->max_grid));
    if (c->trace_on) nchunks = 1; // (BRX_OPTION_TRACE keeps ONE record set of ONE launch: concurrent chunk launches would mix theirs -- ADVICE r4)
    std::vector<uint32_t> cut(nchunks + 1, 0);
    {
        uint32_t i = 0;
        for (unsigned k = 1; k < nchunks; k++) {
            const uint64_t goal = total * k / nchunks;
            while (i < n && (in_off[i] - in_lo) + (out_off[i] - out_lo) < goal) i++;
            cut[k] = i;
        }
        cut[nchunks] = n;
    }
    // Work-queue order inside a chunk: longest compressed stream first.  A stream is a serial job
    // on one wavefront, so a chunk finishes when its longest job does -- start those first.
    std::vector<uint32_t> order(n);
    for (unsigned k = 0; k < nchunks; k++) {
        std::iota(order.begin() + cut[k], order.begin() + cut[k + 1], 0u); // indices are relative to the chunk's first stream
        if (!c->no_order)
            std::stable_sort(order.begin() + cut[k], order.begin() + cut[k + 1], [&](uint32_t x, uint32_t y) {
                const uint32_t gx = cut[k] + x, gy = cut[k] + y;
                return in_off[gx + 1] - in_off[gx] > in_off[gy + 1] - in_off[gy];
            });
    }
    if ((rc = ensure_pool(c, pool_need(c, n < c->max_grid ? n : c->max_grid)))) return rc; // once, for all the chunks' launches together
    // Output buffer in pinned, mapped host memory (brx_host_alloc, hipHostMalloc, hipHostRegister) at the 16-byte phase of
    // the staging slots: the kernel stores every output byte to it as well (BrxKernelArgs::out_mirror) and there is no
    // device-to-host copy of the data afterwards -- it rode on the decode.  The HBM copy stays: it is the window.
    // device-visible address of a host range that is pinned and mapped as ONE piece, else nullptr
    auto mapped = [](const uint8_t *p, size_t bytes) -> uint8_t * {
        hipPointerAttribute_t at;
        void *dp = nullptr, *dp_end = nullptr;
        uint8_t *r = nullptr;
        if (bytes && hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost &&
            hipHostGetDevicePointer(&dp, (void *)p, 0) == hipSuccess && dp &&
            hipHostGetDevicePointer(&dp_end, (void *)(p + bytes - 1), 0) == hipSuccess && (uint8_t *)dp_end == (uint8_t *)dp + bytes - 1)
            r = (uint8_t *)dp;
        (void)hipGetLastError(); // (a pageable pointer makes the queries fail: not an error of this call)
        return r;
    };
    uint8_t *mirror = nullptr;
    if (!c->no_mirror && (((uintptr_t)(out + out_lo)) & 15u) == 0) mirror = mapped(out + out_lo, out_bytes);
    // Compressed input in pinned, mapped host memory: the kernel reads it in place.  A wave stages 256 bytes of input one
    // chunk ahead of its cursor and needs the next one ~40 us later (25 bits per command, ~0.5 us per command): the PCIe
    // round trip hides behind that, 205 MB per 11 ms is a third of the link, and the batch starts decoding at once
    // instead of after its own copy.
    const uint8_t *in_dev = c->no_mirror ? nullptr : mapped(in + in_lo, in_bytes);
    if (!in_dev && (rc = grow(&c->st_in, &c->st_in_cap, in_bytes + 16))) return rc;
    hipStream_t s0 = c->s_chunk[0];
    HIP_TRY(hipMemcpyAsync(d_order, order.data(), (size_t)n * 4, hipMemcpyHostToDevice, s0));
    HIP_TRY(hipMemcpyAsync(d_in_off, hmeta.data(), hmeta.size() * 8, hipMemcpyHostToDevice, s0));
    HIP_TRY(hipEventRecord(c->ev_in[0], s0)); // the tables are on the device
    if (timing) HIP_TRY(hipEventRecord(c->ev[0], s0));
    for (unsigned k = 0; k < nchunks; k++) { // copy in + decode, every chunk on its own stream
        const uint32_t a = cut[k], b = cut[k + 1];
        if (a == b) continue;
        hipStream_t sk = c->s_chunk[k];
        if (k) HIP_TRY(hipStreamWaitEvent(sk, c->ev_in[0], 0));
        const uint64_t i0 = in_off[a] - in_lo, i1 = in_off[b] - in_lo;
        if (i1 > i0 && !in_dev) HIP_TRY(hipMemcpyAsync(c->st_in + i0, in + in_lo + i0, (size_t)(i1 - i0), hipMemcpyHostToDevice, sk));
        uint32_t n_large = 0xffffffffu; // sorted longest first: the streams for the lean instance are the order's tail
        if (!c->no_order) {
            n_large = 0;
            for (uint32_t i = a; i < b; i++) n_large += in_off[i + 1] - in_off[i] > (uint64_t)c->small_bytes ? 1u : 0u;
        }
        rc = launch(c, sk, timing && nchunks == 1, in_dev ? in_dev : c->st_in, d_in_off + a, b - a, c->st_out, d_out_off + a, d_out_len + a,
                    d_status + a, d_order + a, nullptr, nullptr, mirror, nchunks == 1, n_large);
        if (rc) return rc;
    }
    for (unsigned k = 0; k < nchunks; k++) { // copy out (a second loop: a pageable copy blocks the host until it is done)
        const uint32_t a = cut[k], b = cut[k + 1];
        if (a == b) continue;
        hipStream_t sk = c->s_chunk[k];
        const uint64_t o0 = out_off[a] - out_lo, o1 = out_off[b] - out_lo;
        if (o1 > o0 && !mirror) HIP_TRY(hipMemcpyAsync(out + out_lo + o0, c->st_out + o0, (size_t)(o1 - o0), hipMemcpyDeviceToHost, sk));
        HIP_TRY(hipMemcpyAsync(out_len + a, d_out_len + a, (size_t)(b - a) * 8, hipMemcpyDeviceToHost, sk));
        HIP_TRY(hipMemcpyAsync(status + a, d_status + a, (size_t)(b - a) * 4, hipMemcpyDeviceToHost, sk));
    }
    if (timing) HIP_TRY(hipEventRecord(c->ev[1], s0));
    for (unsigned k = 0; k < nchunks; k++) HIP_TRY(hipStreamSynchronize(c->s_chunk[k]));
    c->have_timing = timing && nchunks == 1;
    return BRX_SUCCESS;
}

static int decode_batch_locked(brx_ctx *c, const uint8_t *in, const uint64_t *in_off, uint32_t n, uint8_t *out,
                               const uint64_t *out_off, uint64_t *out_len, int32_t *status, const brx_opts *opts) {
    const uint32_t flags = opts ? opts->flags : 0u;
    const bool timing = (flags & BRX_OPT_TIMING) != 0;
    HIP_TRY(hipSetDevice(c->device));
    c->have_timing = false;
    if (flags & BRX_MEM_DEVICE) {
        hipStream_t st = (opts && opts->hip_stream) ? (hipStream_t)opts->hip_stream : c->stream;
        const uint32_t *d_order = nullptr;
        uint32_t n_large = 0xffffffffu;
        if (flags & BRX_OPT_ORDER) { // longest compressed stream first, from a copy of the device table
            std::vector<uint64_t> h((size_t)n + 1);
            HIP_TRY(hipMemcpyAsync(h.data(), in_off, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            std::vector<uint32_t> order(n);
            std::iota(order.begin(), order.end(), 0u);
            std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return h[x + 1] - h[x] > h[y + 1] - h[y]; });
            int rc0 = ensure_order(c, n);
            if (rc0) return rc0;
            uint32_t *slot = c->d_order + (size_t)(c->launch_seq % BRX_COUNTER_RING) * c->order_cap; // the slot launch() takes next
            HIP_TRY(hipMemcpyAsync(slot, order.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st)); // (`order` is a local)
            d_order = slot;
            n_large = 0;
            for (uint32_t i = 0; i < n; i++) n_large += h[i + 1] - h[i] > (uint64_t)c->small_bytes ? 1u : 0u;
        }
        if (timing) HIP_TRY(hipEventRecord(c->ev[0], st));
        int rc = launch(c, st, timing, in, in_off, n, out, out_off, out_len, status, d_order, nullptr, nullptr, nullptr, true, n_large);
        if (rc) return rc;
        if (timing) HIP_TRY(hipEventRecord(c->ev[1], st));
        if (!(opts && opts->hip_stream)) HIP_TRY(hipStreamSynchronize(st));
        c->have_timing = timing;
        return BRX_SUCCESS;
    }
    return decode_host(c, in, in_off, n, out, out_off, out_len, status, timing);
}

extern "C" int brx_decode_batch(brx_ctx *c, const uint8_t *in, const uint64_t *in_off, uint32_t n, uint8_t *out,
                                const uint64_t *out_off, uint64_t *out_len, int32_t *status, const brx_opts *opts) {
    BRX_GUARD_BEGIN
    if (!c) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_decode_batch: ctx is NULL");
    if (n == 0) return BRX_SUCCESS;
    if (!in_off || !out_off || !out_len || !status) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_decode_batch: NULL table");
    std::lock_guard<std::mutex> lk(c->mu);
    return decode_batch_locked(c, in, in_off, n, out, out_off, out_len, status, opts);
    BRX_GUARD_END(BRX_ERR_OUT_OF_MEMORY, BRX_ERR_HIP)
}

extern "C" double brx_last_timing(brx_ctx *c, int which) {
    if (!c) return -1.0;
    std::lock_guard<std::mutex> lk(c->mu);
    if (which == 12 || which == 13) { // 12: waves that ever had to wait for a spill slab (0 by construction: pool_need); 13: slabs of the pool
        if (which == 13) return (double)c->pool.count;
        uint32_t w = 0;
        if (c->pool.waits) {
            if (hipSetDevice(c->device) != hipSuccess) return -1.0;
            if (c->any_launch && hipEventSynchronize(c->ev_last) != hipSuccess) return -1.0;
            if (hipMemcpy(&w, c->pool.waits, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1.0;
        }
        return (double)(c->slab_waits + w);
    }
    if (which == 14) return (double)c->facade_batches; // the Read facade: batches launched for queued streams ...
    if (which == 15) return (double)c->facade_streams; // ... and the streams in them (status-25 retries count again)
    if (which == 9) return (double)c->stream_regrown; // bounded streams of this context: pauses in front of one item that needed more room behind the window
    if (which == 8) return (double)c->stream_short_slices; // bounded streams of this context: slices that paused in front of an item the resident input did not hold
    if ((which >= 2 && which <= 7) || which == 10 || which == 11) { // counters of the most recent launch (waits for it): 2..4 = streams decoded at level >= which - 1
                                    // (2: every stream that left the regular kernel); 5 = streams the lean instance left to the
                                    // regular kernel (large ones + given up); 6 = streams of the late list (handed up with their
                                    // state at a later meta-block); 7 = bytes decoded twice (0: every one of those was resumed)
        if (!c->any_launch) return -1.0;
        if (!c->last_counter) return 0.0;
        uint32_t w[32];
        if (hipEventSynchronize(c->ev_last) != hipSuccess) return -1.0;
        if (hipMemcpy(w, c->last_counter, 128, hipMemcpyDeviceToHost) != hipSuccess) return -1.0;
        if (which == 10) return (double)w[18]; // meta-blocks taken back after a speculative end (the stream read on past its input)
        if (which == 11) return (double)std::min<uint32_t>(w[19], BRX_LATE2_CAP); // streams level 3 handed on to level 4
        const uint32_t late = std::min<uint32_t>(w[8], BRX_LATE_CAP);
        switch (which) {
        case 2: return (double)w[5] + w[6] + w[7] + late;
        case 3: return (double)w[6] + w[7];
        case 4: return (double)w[7];
        case 5: return (double)w[10];
        case 6: return (double)late;
        default: return (double)w[11];
        }
    }
    if (!c->have_timing) return -1.0;
    float ms = 0.f;
    hipError_t e = which == 0 ? hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) : hipEventElapsedTime(&ms, c->ev[2], c->ev[3]);
    return e == hipSuccess ? (double)ms : -1.0;
}

extern "C" int brx_last_trace(brx_ctx *c, uint64_t *dst, uint32_t n) {
    if (!c || !dst) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_last_trace: NULL argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->d_trace || n > c->trace_n) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_last_trace: no trace of that many streams (BRX_OPTION_TRACE)");
    HIP_TRY(hipSetDevice(c->device));
    if (c->any_launch) HIP_TRY(hipEventSynchronize(c->ev_last));
    HIP_TRY(hipMemcpy(dst, c->d_trace, (size_t)n * 32u, hipMemcpyDeviceToHost));
    return BRX_SUCCESS;
}

extern "C" int brx_synchronize(brx_ctx *c, void *hip_stream) {
    if (!c) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_synchronize: ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(hip_stream ? (hipStream_t)hip_stream : c->stream));
    return BRX_SUCCESS;
}

// ---- stream generator (brx_gen.hip): n inputs -> n valid Brotli streams, made on the device ------------------------------
void brx_launch_generate(const void *src, const uint64_t *src_off, uint32_t n, void *out, const uint64_t *out_off,
                         uint64_t *out_len, int32_t *status, const void *header, uint32_t header_bits, uint32_t mb_bytes,
                         uint32_t switches, uint32_t *hash, void *hip_stream);
void brx_launch_generate_adaptive(const void *src, const uint64_t *src_off, uint32_t n, void *out, const uint64_t *out_off,
                                  uint64_t *out_len, int32_t *status, uint32_t mb_bytes, uint32_t *hash, void *cmds, uint32_t cmd_cap,
                                  const void *context_lut, void *hip_stream);

extern "C" int brx_generate_batch(brx_ctx *c, const uint8_t *src, const uint64_t *src_off, uint32_t n, uint8_t *out,
                                  const uint64_t *out_off, uint64_t *out_len, int32_t *status, uint32_t metablock_bytes,
                                  const brx_opts *opts) {
    BRX_GUARD_BEGIN
    if (!c) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_generate_batch: ctx is NULL");
    if (n == 0) return BRX_SUCCESS;
    if (!src_off || !out_off || !out_len || !status) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_generate_batch: NULL table");
    if (metablock_bytes == 0) metablock_bytes = 65536;
    if (metablock_bytes > (1u << 24)) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_generate_batch: a meta-block holds at most 2^24 bytes");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t flags = opts ? opts->flags : 0u;
    // tables/gen_header.bin: record A (one literal block type), record B (two, switching): u32 bits + bits padded to 4 bytes
    auto rd32 = [](const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); };
    const uint32_t bits_a = rd32(BRX_GEN_HEADER);
    const size_t rec_b = 4 + (((size_t)bits_a + 7) / 8 + 3) / 4 * 4;
    const uint32_t switches = (flags & BRX_GEN_SWITCHES) ? 1u : 0u;
    const uint32_t header_bits = switches ? rd32(BRX_GEN_HEADER + rec_b) : bits_a;
    const size_t header_at = (switches ? rec_b : 0) + 4;
    if (!c->d_gen_header) {
        HIP_TRY(hipMalloc(&c->d_gen_header, sizeof BRX_GEN_HEADER));
        HIP_TRY(hipMemcpy(c->d_gen_header, BRX_GEN_HEADER, sizeof BRX_GEN_HEADER, hipMemcpyHostToDevice));
    }
    // BRX_GEN_ADAPTIVE: one wavefront per stream, codes from the meta-block's own statistics; the commands of the meta-block
    // being written wait in a scratch list (12 bytes each, at most one per 4 input bytes)
    hipStream_t st_for_run = (opts && opts->hip_stream && (flags & BRX_MEM_DEVICE)) ? (hipStream_t)opts->hip_stream : c->stream;
    const bool adaptive = (flags & BRX_GEN_ADAPTIVE) != 0;
    const uint32_t cmd_cap = metablock_bytes / 4u + 2u;
    const size_t per_stream = 2048u * 4u + (adaptive ? (size_t)cmd_cap * 12u : 0u);
    // streams per launch: bounds the scratch (hash tables, 8 KiB per stream; command lists) at ~1 GiB
    const uint32_t per_launch = (uint32_t)std::min<size_t>(32768, std::max<size_t>(1, ((size_t)1 << 30) / per_stream)); // (no floor above 1: ADVICE r3 -- 64 streams of 2^24-byte meta-blocks would ask for 3.2 GB)
    const size_t hash_bytes = (size_t)std::min(n, per_launch) * per_stream;
    auto run = [&](const void *s_, const uint64_t *so, uint32_t m, void *o_, const uint64_t *oo, uint64_t *ol, int32_t *stt) {
        if (adaptive)
            brx_launch_generate_adaptive(s_, so, m, o_, oo, ol, stt, metablock_bytes, (uint32_t *)c->st_gen,
                                         c->st_gen + (size_t)std::min(n, per_launch) * 2048u * 4u, cmd_cap, c->d_lut, st_for_run);
        else
            brx_launch_generate(s_, so, m, o_, oo, ol, stt, c->d_gen_header + header_at, header_bits, metablock_bytes, switches,
                                (uint32_t *)c->st_gen, st_for_run);
    };
    hipStream_t st = (opts && opts->hip_stream && (flags & BRX_MEM_DEVICE)) ? (hipStream_t)opts->hip_stream : c->stream;
    if (flags & BRX_MEM_DEVICE) {
        // (device-pointer calls are asynchronous on the caller's stream and share this scratch: before it is REPLACED by a larger
        // one nobody may be using the old one any more -- ADVICE r3; calls that fit run one behind the other on their streams'
        // order only if they use the same stream: concurrent generator calls on different streams must use different contexts)
        if (hash_bytes > c->st_gen_cap || !c->st_gen) HIP_TRY(hipDeviceSynchronize());
        int rc = grow(&c->st_gen, &c->st_gen_cap, hash_bytes);
        if (rc) return rc;
        for (uint32_t k = 0; k < n; k += per_launch) { // (same stream: the launches run one after the other and share the tables)
            const uint32_t m = std::min(per_launch, n - k);
            run(src, src_off + k, m, out, out_off + k, out_len + k, status + k);
            HIP_TRY(hipGetLastError());
        }
        if (!(opts && opts->hip_stream)) HIP_TRY(hipStreamSynchronize(st));
        return BRX_SUCCESS;
    }
    // host pointers: stage in, generate, stage out
    for (uint32_t i = 0; i < n; i++)
        if (src_off[i + 1] < src_off[i] || out_off[i + 1] < out_off[i])
            return fail(BRX_ERR_INVALID_ARGUMENT, "brx_generate_batch: offsets must be non-decreasing");
    const uint64_t s_lo = src_off[0], o_lo = out_off[0];
    const size_t s_bytes = (size_t)(src_off[n] - s_lo), o_bytes = (size_t)(out_off[n] - o_lo), tab = ((size_t)n + 1) * 8;
    if ((s_bytes && !src) || (o_bytes && !out)) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_generate_batch: NULL data");
    const size_t a_src = (hash_bytes + 255) & ~(size_t)255, a_out = (a_src + s_bytes + 255) & ~(size_t)255,
                 a_tab = (a_out + o_bytes + 255) & ~(size_t)255, total = a_tab + 3 * tab + (size_t)n * 4 + 64;
    int rc = grow(&c->st_gen, &c->st_gen_cap, total);
    if (rc) return rc;
    std::vector<uint64_t> h(2 * ((size_t)n + 1));
    for (uint32_t i = 0; i <= n; i++) { h[i] = src_off[i] - s_lo; h[(size_t)n + 1 + i] = out_off[i] - o_lo; }
    uint64_t *d_soff = (uint64_t *)(c->st_gen + a_tab), *d_ooff = d_soff + (n + 1), *d_len = d_ooff + (n + 1);
    int32_t *d_st = (int32_t *)(d_len + (n + 1));
    if (s_bytes) HIP_TRY(hipMemcpyAsync(c->st_gen + a_src, src + s_lo, s_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_soff, h.data(), 2 * tab, hipMemcpyHostToDevice, st));
    for (uint32_t k = 0; k < n; k += per_launch) {
        const uint32_t m = std::min(per_launch, n - k);
        run(c->st_gen + a_src, d_soff + k, m, c->st_gen + a_out, d_ooff + k, d_len + k, d_st + k);
        HIP_TRY(hipGetLastError());
    }
    if (o_bytes) HIP_TRY(hipMemcpyAsync(out + o_lo, c->st_gen + a_out, o_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_len, d_len, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(status, d_st, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return BRX_SUCCESS;
    BRX_GUARD_END(BRX_ERR_OUT_OF_MEMORY, BRX_ERR_HIP)
}

// ---- compaction of a decoded batch (brx_util.hip) ------------------------------------------------------------------------
void brx_launch_compact(const void *src, const uint64_t *src_off, const uint64_t *len, void *dst, const uint64_t *dst_off,
                        uint32_t n, uint64_t total, void *hip_stream);

extern "C" int brx_compact_batch(brx_ctx *c, const uint8_t *out, const uint64_t *out_off, const uint64_t *len, uint32_t n,
                                 uint8_t *dst, const uint64_t *dst_off, uint64_t total, void *hip_stream) {
    BRX_GUARD_BEGIN
    if (!c) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_compact_batch: ctx is NULL");
    if (n == 0 || total == 0) return BRX_SUCCESS;
    if (!out || !out_off || !len || !dst || !dst_off) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_compact_batch: NULL argument");
    if ((total + 16383u) / 16384u > 0x7fffffffull) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_compact_batch: more than 32 TiB");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    brx_launch_compact(out, out_off, len, dst, dst_off, n, total, st);
    HIP_TRY(hipGetLastError());
    if (!hip_stream) HIP_TRY(hipStreamSynchronize(st));
    return BRX_SUCCESS;
    BRX_GUARD_END(BRX_ERR_OUT_OF_MEMORY, BRX_ERR_HIP)
}

extern "C" void *brx_host_alloc(size_t bytes) {
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        fail(BRX_ERR_OUT_OF_MEMORY, "hipHostMalloc", e);
        return nullptr;
    }
    return p;
}

extern "C" void brx_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

// ---- Read-shaped facade: one object = one stream (reference Decompressor<R>, src/lib.rs:377-410, 2173-2193) ----------
// Streams created on one context and not yet read are decoded TOGETHER, as one batch, by the first read of any of
// them (256 live Decompressors cost about one batch, not 256 launches).  A stream keeps the bytes produced before an
// error and serves them first, like the reference does.
struct brx_stream {
    brx_ctx *ctx;
    std::vector<uint8_t> in;
    std::vector<uint8_t> out;
    size_t served = 0;
    std::atomic<bool> decoded{false}; // (set by whichever thread led the batch; read without a lock by the stream's owner)
    std::condition_variable cv;       // its owner waits here while another reader's batch carries it (or runs ahead of it)
    bool waiting = false;             // ... and is listed in brx_ctx::waitq (both under the queue's lock)
    const uint8_t *view = nullptr;    // its bytes still in the context's staging: the owner (blocked in brx_stream_read when the batch ended)
    size_t view_len = 0;              // copies them out itself -- N owners copy at once instead of the leading thread N times
    int32_t status = 0;
    int lib_rc = BRX_SUCCESS;
    // bounded mode (large streams, and every stream over a reader): decoded slice by slice into a sliding device window of
    // output, from a sliding device window of compressed input -- never more resident than BRX_BOUNDED_BUFSIZE of output +
    // the input window (+ one spill slab and one state record) on the device, one staging chunk on the host
    bool bounded = false, finished = false;
    brx_read_fn read_fn = nullptr; // the compressed input is PULLED (reference: BufReader over R, src/bitreader/mod.rs:21-53);
    void *read_user = nullptr;     // nullptr = from `in` (brx_stream_new / brx_stream_new_bounded)
    size_t mem_at = 0;             // ... how much of `in` has been pulled
    bool src_eof = false;
    bool no_progress = false, stalled = false; // the last slice paused where it started / ... and the window could not be improved
    bool pull_broken = false;                  // the read callback returned more than it was given room for
    size_t buf_size = 0;                       // size of d_buf: BRX_BOUNDED_BUFSIZE, more once a single command needed more
    uint64_t want_room = 0;                    // the output position the item in front of the last pause runs to (BrxResume::need_room), or 0
    bool want_room_optional = false;           // ... it is a whole META-BLOCK (round 6): without the room it still decodes, in the slower loop
    bool no_mb_room = false;                   // that room could not be had once: the kernel is told not to pause in front of meta-blocks any more
    uint8_t *d_inwin = nullptr, *d_buf = nullptr;
    size_t in_window = (8u << 20);     // size of d_inwin (the context's reader_window when the stream started decoding)
    size_t in_fill = 0, in_cursor = 0; // bytes resident in d_inwin; the decoder's cursor in it (after the last good slice)
    uint64_t in_slide_pending = 0;     // bytes the window has moved up since the record was last told
    std::vector<uint8_t> stage;
    BrxResume *d_rec = nullptr;
    BrxSlabPool *d_pool = nullptr;
    uint32_t *d_bitmap = nullptr, *d_slab = nullptr;
    uint64_t *d_meta = nullptr; // in_off[2] | out_off[2] | out_len[1] | status
    uint64_t shift = 0, pos = 0, delivered = 0;
};
#define BRX_BOUNDED_WINDOW (16u << 20) // the largest Brotli window, (1 << 24) - 16, rounded up
#define BRX_BOUNDED_CHUNK (4u << 20)   // output decoded per slice
#define BRX_BOUNDED_SLACK ((1u << 20) + 65536u) // room for the command that crosses the slice end
#define BRX_FACADE_LINGER_QUIET_US 120         // the leading reader goes once no stream has been queued for this long ...
#define BRX_FACADE_LINGER_MAX_US 1500          // ... or after this long
#define BRX_FACADE_LINGER_BELOW 96             // ... and only waits at all while batches are smaller than this
#define BRX_FACADE_PINNED_MAX ((size_t)1 << 30) // pinned staging the Read facade keeps per context, at most
#define BRX_BOUNDED_THRESHOLD (4u << 20)        // brx_stream_new: compressed inputs from this size on are decoded bounded
#define BRX_BOUNDED_SLIDE_MIN (1u << 20)        // the window slides only once it is over by this much (see bounded_step)
#define BRX_BOUNDED_BUFSIZE ((size_t)BRX_BOUNDED_WINDOW + BRX_BOUNDED_SLIDE_MIN + BRX_BOUNDED_CHUNK + BRX_BOUNDED_SLACK)
#define BRX_IN_WINDOW (8u << 20)      // compressed bytes resident on the device (default; BRX_OPTION_READER_WINDOW)
#define BRX_IN_KEEP 4096u             // ... of which this much below the cursor stays (the kernel stages 256-byte chunks behind it)
#define BRX_IN_STAGE (1u << 20)       // host staging chunk of the pulls
#define BRX_IN_MARGIN_DIV 32u         // a slice pauses window / 32 (256 KiB) in front of the resident end while the source has more
#define BRX_IN_WINDOW_MAX ((size_t)256u << 20) // the input window grows up to this when one item needs more than it holds
#define BRX_BOUNDED_BUF_MAX ((size_t)BRX_BOUNDED_BUFSIZE + ((size_t)48u << 20)) // ... the output buffer when one command produces more than the slack
                                      // (an insert or a copy is < 2^24 + 2^22 bytes each, an uncompressed meta-block <= 2^24)

static void bounded_release(brx_stream *s) {
    if (!s->d_buf && !s->d_inwin) return;
    if (s->ctx) (void)hipSetDevice(s->ctx->device);
    (void)hipFree(s->d_inwin);
    (void)hipFree(s->d_buf);
    (void)hipFree(s->d_slab);
    (void)hipFree(s->d_rec); // (one allocation: the record, the pool descriptor, the bitmap, the offset / result words)
    s->d_inwin = s->d_buf = nullptr;
    s->d_rec = nullptr;
    s->d_pool = nullptr;
    s->d_bitmap = s->d_slab = nullptr;
    s->d_meta = nullptr;
    std::vector<uint8_t>().swap(s->stage);
}

static int bounded_init(brx_stream *s) {
    brx_ctx *c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    s->in_window = c->reader_window;
    s->no_mb_room = !c->reader_mb_room;
    s->buf_size = BRX_BOUNDED_BUFSIZE;
    HIP_TRY(hipMalloc(&s->d_inwin, s->in_window + 16)); // (the option's size, not the default's: ADVICE r4)
    HIP_TRY(hipMalloc(&s->d_buf, s->buf_size));
    HIP_TRY(hipMalloc(&s->d_slab, (size_t)BRX_SCRATCH_WORDS * 4u));
    const size_t rec_bytes = (sizeof(BrxResume) + 255u) & ~(size_t)255;
    uint8_t *small = nullptr;
    HIP_TRY(hipMalloc(&small, rec_bytes + 1024));
    s->d_rec = (BrxResume *)small;
    s->d_pool = (BrxSlabPool *)(small + rec_bytes);
    s->d_bitmap = (uint32_t *)(small + rec_bytes + 256);
    s->d_meta = (uint64_t *)(small + rec_bytes + 512);
    HIP_TRY(hipMemset(s->d_rec, 0, 32));
    const uint32_t only_slab_0 = 0xfffffffeu; // a private pool of ONE slab that survives between the slices
    HIP_TRY(hipMemcpy(s->d_bitmap, &only_slab_0, 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(s->d_bitmap + 1, 0, 4));
    BrxSlabPool p = {s->d_bitmap, s->d_slab, 32, s->d_slab, s->d_bitmap + 1}; // (sink = the one slab: only this stream's one wave ever asks)
    HIP_TRY(hipMemcpy(s->d_pool, &p, sizeof p, hipMemcpyHostToDevice));
    s->stage.resize(BRX_IN_STAGE);
    return BRX_SUCCESS;
}

// Pull compressed bytes: up to `cap` into buf, 0 = the source is exhausted.  A pulled source is the caller's code: it runs with
// the context's lock RELEASED -- it may itself read from another brx_stream / Decompressor of the same context (nested readers;
// with the lock held that thread would wait for itself) -- and what it returns is checked against the room it was given.
static size_t stream_pull(brx_stream *s, uint8_t *buf, size_t cap, std::unique_lock<std::mutex> &lk) {
    if (s->read_fn) {
        lk.unlock();
        size_t got = 0;
        try {
            got = s->read_fn(s->read_user, buf, cap);
        } catch (...) {
            lk.lock();
            throw;
        }
        lk.lock();
        // (the callback may have been a reader on a context of another device, and another thread may have destroyed this stream's
        // context meanwhile -- against the contract, but cheap to notice: ADVICE r5)
        if (s->ctx == nullptr || s->d_inwin == nullptr || hipSetDevice(s->ctx->device) != hipSuccess) {
            s->pull_broken = true;
            return 0;
        }
        if (got > cap) {
            s->pull_broken = true;
            return 0;
        }
        return got;
    }
    const size_t k = std::min(cap, s->in.size() - s->mem_at);
    if (k) memcpy(buf, s->in.data() + s->mem_at, k);
    s->mem_at += k;
    return k;
}

// Move the input window up to the cursor (when that frees at least 1 MiB) and fill it from the source.
static int bounded_refill(brx_stream *s, std::unique_lock<std::mutex> &lk) {
    brx_ctx *c = s->ctx;
    if (s->in_cursor >= BRX_IN_KEEP + s->in_window / 8u) {
        const size_t delta = (s->in_cursor - BRX_IN_KEEP) & ~(size_t)15, keep = s->in_fill - delta;
        for (size_t done = 0; done < keep; done += delta) { // forward, in pieces no longer than the move: no overlap
            const size_t piece = std::min(delta, keep - done);
            HIP_TRY(hipMemcpyAsync(s->d_inwin + done, s->d_inwin + delta + done, piece, hipMemcpyDeviceToDevice, c->stream));
        }
        s->in_fill -= delta;
        s->in_cursor -= delta;
        s->in_slide_pending += delta;
    }
    while (!s->src_eof && s->in_fill < s->in_window) {
        const size_t got = stream_pull(s, s->stage.data(), std::min<size_t>(s->stage.size(), s->in_window - s->in_fill), lk);
        if (s->pull_broken) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_stream_read: the read callback returned more bytes than it was given room for");
        if (got == 0) {
            s->src_eof = true;
            break;
        }
        HIP_TRY(hipMemcpyAsync(s->d_inwin + s->in_fill, s->stage.data(), got, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream)); // (the staging chunk is reused)
        s->in_fill += got;
    }
    return BRX_SUCCESS;
}

// One item (a header, an uncompressed meta-block, one command) needs more compressed bytes than the input window holds: a
// window twice the size, same content.  (Before round 5 such a slice ran with "this is all there is" and reported UnexpectedEOF.)
static int bounded_grow_in(brx_stream *s) {
    brx_ctx *c = s->ctx;
    if (s->in_window >= BRX_IN_WINDOW_MAX)
        return fail(BRX_ERR_OUT_OF_MEMORY, "brx_stream_read: one item of the stream needs more than 256 MiB of compressed input resident");
    const size_t bigger = std::min(s->in_window * 2u, BRX_IN_WINDOW_MAX);
    uint8_t *nb = nullptr;
    hipError_t e = hipMalloc(&nb, bigger + 16);
    if (e != hipSuccess) return fail(BRX_ERR_OUT_OF_MEMORY, "brx_stream_read: input window allocation failed", e);
    if (hipMemcpyAsync(nb, s->d_inwin, s->in_fill, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        (void)hipFree(nb);
        return fail(BRX_ERR_HIP, "brx_stream_read: input window copy failed");
    }
    (void)hipFree(s->d_inwin);
    s->d_inwin = nb;
    s->in_window = bigger;
    return BRX_SUCCESS;
}

// One command produces more than the output buffer has room for behind the window (a long copy, a long insert, an uncompressed
// meta-block): a buffer that holds it, same content.  `need_abs` = the stream position the command runs to.
static int bounded_grow_out(brx_stream *s, uint64_t need_abs) {
    brx_ctx *c = s->ctx;
    if (need_abs <= s->shift) return BRX_ERR_OUT_OF_MEMORY;
    size_t want = (size_t)(need_abs - s->shift) + BRX_BOUNDED_SLACK;
    want = (want + 0xfffffu) & ~(size_t)0xfffffu;
    if (want <= s->buf_size || want > BRX_BOUNDED_BUF_MAX) return BRX_ERR_OUT_OF_MEMORY;
    uint8_t *nb = nullptr;
    hipError_t e = hipMalloc(&nb, want);
    if (e != hipSuccess) return fail(BRX_ERR_OUT_OF_MEMORY, "brx_stream_read: output window allocation failed", e);
    if (hipMemcpyAsync(nb, s->d_buf, (size_t)(s->pos - s->shift), hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        (void)hipFree(nb);
        return fail(BRX_ERR_HIP, "brx_stream_read: output window copy failed");
    }
    (void)hipFree(s->d_buf);
    s->d_buf = nb;
    s->buf_size = want;
    return BRX_SUCCESS;
}

// Decode the next slice: up to BRX_BOUNDED_CHUNK more bytes (to the next command boundary past it).
static int bounded_step(brx_stream *s, std::unique_lock<std::mutex> &lk) {
    brx_ctx *c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    if (s->pos - s->shift >= (uint64_t)BRX_BOUNDED_WINDOW + BRX_BOUNDED_SLIDE_MIN) {
        // slide the window: keep the last BRX_BOUNDED_WINDOW bytes (every back-reference reaches at most that far); the
        // base moves by a multiple of 16 so the kernel's 16-byte store alignment (its ring skew) is unchanged.  The move is
        // an overlapping one, done forward in pieces no longer than the distance moved -- so the window slides only once it
        // is over by BRX_BOUNDED_SLIDE_MIN (the buffer has that much more room): at most ~17 copies per slide.  (The kernel
        // pauses at an arbitrary command boundary: sliding as soon as the window was over by 16 bytes made the FIRST slide of
        // every stream a move of 16 MiB in 16 .. 200-byte pieces, 10^5 copies -- ADVICE r3; over by 1..15 bytes it never ended,
        // ADVICE r2.)
        const uint64_t new_shift = (s->pos - BRX_BOUNDED_WINDOW) & ~15ull;
        const uint64_t delta = new_shift - s->shift, keep = s->pos - new_shift; // delta >= BRX_BOUNDED_SLIDE_MIN - 15
        for (uint64_t done = 0; done < keep; done += delta) { // forward, in pieces no longer than the move: no overlap
            const size_t piece = (size_t)std::min<uint64_t>(delta, keep - done);
            HIP_TRY(hipMemcpyAsync(s->d_buf + done, s->d_buf + delta + done, piece, hipMemcpyDeviceToDevice, c->stream));
        }
        s->shift = new_shift;
    }
    if (s->want_room > BRX_STREAM_LIMIT) return BRX_ERR_OUT_OF_MEMORY; // (the stream runs past the 4 GiB - 256 B of the position arithmetic)
    if (s->want_room > s->shift + s->buf_size) { // (the item the last slice paused in front of: still beyond the capacity after the slide)
        int rc = bounded_grow_out(s, s->want_room);
        if (rc && s->want_room_optional) s->no_mb_room = true; // (a meta-block decodes without: one command per call, each checked against the capacity)
        else if (rc) return BRX_ERR_OUT_OF_MEMORY;
    }
    s->want_room = 0;
    // the input side: keep at least half a window of compressed bytes in front of the cursor while the source has any
    const uint64_t pos0 = s->pos;
    if (!s->src_eof && (s->in_fill - s->in_cursor < s->in_window / 2u || s->no_progress)) {
        const size_t fill0 = s->in_fill; // (a slice that paused at its in_low comes here too)
        const uint64_t slide0 = s->in_slide_pending;
        int rc = bounded_refill(s, lk);
        if (rc) return rc;
        if (s->no_progress && !s->src_eof && s->in_fill == fill0 && s->in_slide_pending == slide0) {
            // the slice paused where it had started and the window can neither move nor take more: ONE item needs more input
            // than the window holds -- a bigger window (only when that fails does the slice run as if this were all there is)
            rc = bounded_grow_in(s);
            if (rc == BRX_SUCCESS) rc = bounded_refill(s, lk);
            else if (rc == BRX_ERR_OUT_OF_MEMORY) { // (the window is at its limit, or the allocation failed: this slice is told that what is
                s->stalled = true;                   // resident is all there is and reports the status it finds -- the reference's
                rc = BRX_SUCCESS;                    // UnexpectedEOF, not a library error: ADVICE r5)
            }
            if (rc) return rc;
        }
    }
    s->no_progress = false;
    const uint64_t cap_abs = std::min<uint64_t>(s->shift + s->buf_size, BRX_STREAM_LIMIT);
    const uint64_t pause_at = s->pos + BRX_BOUNDED_CHUNK;
    uint8_t *virt = (uint8_t *)((uintptr_t)s->d_buf - (uintptr_t)s->shift); // address of output byte 0, were it still resident
    {
        uint64_t meta[4] = {0, s->in_fill, 0, cap_abs};
        HIP_TRY(hipMemcpyAsync(s->d_meta, meta, sizeof meta, hipMemcpyHostToDevice, c->stream));
        // While the source has more, the slice pauses a margin (1/32 of the window: 256 KiB) short of the resident end -- at a
        // command or meta-block boundary, or in the middle of a literal run -- and a segment that still runs into that end (a
        // header, an uncompressed block, one whole command of the C++ loop) is taken back by the kernel itself: the slice
        // pauses in FRONT of it (brx_kernels.hip, resumable mode).  So UnexpectedEOF comes out of a slice only when it was told
        // that the resident input is all there is: the source is dry -- or `stalled`: the last slice paused where it had started, the
        // window could not be moved or filled any further and could not grow either (one item that needs more than 256 MiB of input
        // resident); this slice reports what it finds.  (UnexpectedEOF that the reference raises for a FORMAT error -- a bad
        // MSKIPLEN, Q10 -- is not taken back by the kernel: ST_EOF_FORMAT in brx_kernels.hip.)
        const size_t margin = s->in_window / BRX_IN_MARGIN_DIV;
        const uint64_t in_low = s->src_eof || s->in_fill <= margin || s->stalled ? ~0ull : 8ull * (s->in_fill - margin);
        const uint64_t pz[4] = {pause_at, s->in_slide_pending, in_low, s->no_mb_room ? 1ull : 0ull}; // (need_room on the way in: 1 = no pause in front of whole meta-blocks)
        HIP_TRY(hipMemcpyAsync((uint8_t *)s->d_rec + offsetof(BrxResume, pause_at), pz, 32, hipMemcpyHostToDevice, c->stream));
        int rc = launch(c, c->stream, false, s->d_inwin, s->d_meta, 1, virt, s->d_meta + 2, s->d_meta + 4,
                        (int32_t *)(s->d_meta + 5), nullptr, s->d_rec, s->d_pool);
        if (rc) return rc;
        uint64_t res[2] = {0, 0}, cur = 0, need_room = 0;
        HIP_TRY(hipMemcpyAsync(&need_room, &s->d_rec->need_room, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(res, s->d_meta + 4, 16, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(&cur, &s->d_rec->lds[BRX_RESUME_CURSOR_WORD], 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        const int32_t st = (int32_t)(res[1] & 0xffffffffu);
        if (st == BRX_OUTPUT_TOO_SMALL) return BRX_ERR_OUT_OF_MEMORY; // (not reached since round 5: the kernel takes such an item back and
                                                                      // pauses in front of it -- need_room below; kept as the caller's fallback)
        s->in_slide_pending = 0;
        s->stalled = false;
        s->pos = res[0];
        if (st != BRX_PAUSED) {
            s->finished = true;
            s->status = st;
            if (st == BRX_OK && !s->src_eof) { // bytes behind the end of the stream (StreamEnd, src/lib.rs:2155-2167)?
                uint8_t probe;
                if (stream_pull(s, &probe, 1, lk) != 0) s->status = BRX_EXPECTED_END_OF_STREAM;
                else s->src_eof = true;
            }
        } else {
            const size_t cursor = (size_t)(cur >> 3);
            // One item (a long copy or insert, an uncompressed meta-block) runs to `need_room` and did not fit behind the window: the
            // kernel took it back and paused in front of it.  The next slice first slides the window; if the item still does not
            // fit, the buffer grows to hold it (bounded_grow_out) -- nothing is decoded twice, nothing is put back.
            s->want_room_optional = (need_room >> 63) != 0;
            need_room &= ~(1ull << 63);
            s->want_room = need_room;
            if (need_room) c->stream_regrown++;
            s->no_progress = cursor == s->in_cursor && res[0] == pos0 && need_room == 0;
            if (res[0] < pause_at && cur < in_low) c->stream_short_slices++; // (paused in front of something that did not fit what was resident)
            s->in_cursor = cursor;
        }
        return BRX_SUCCESS;
    }
}

static void stream_detach(brx_stream *s) {
    bounded_release(s); // (while s->ctx still names the device)
    s->ctx = nullptr;
    if (s->bounded && !s->finished && s->lib_rc == BRX_SUCCESS) s->lib_rc = BRX_ERR_INVALID_ARGUMENT;
    if (!s->decoded) {
        s->decoded = true;
        s->lib_rc = BRX_ERR_INVALID_ARGUMENT; // the context went away before the stream was read
    }
}

// The mode is decided BEFORE the stream becomes visible to the context: a bounded stream never sits in `pending` (a
// concurrent read of another stream would decode it in its batch and empty its input -- ADVICE r2).
static brx_stream *stream_new_impl(brx_ctx *ctx, const uint8_t *in, size_t n, bool force_bounded) {
    BRX_GUARD_BEGIN
    if (!ctx || (n && !in)) {
        fail(BRX_ERR_INVALID_ARGUMENT, "brx_stream_new: bad argument");
        return nullptr;
    }
    brx_stream *s = new brx_stream();
    s->ctx = ctx;
    try {
        s->in.assign(in, in + n);
        s->bounded = force_bounded || n >= BRX_BOUNDED_THRESHOLD;
        std::lock_guard<std::mutex> ql(ctx->qmu); // (never the context's lock: a running batch does not hold up the making of streams)
        ctx->live.push_back(s);
        if (!s->bounded) {
            ctx->pending.push_back(s);
            if (ctx->facade_busy) ctx->new_cv.notify_one();
        }
    } catch (...) {
        try {
            std::lock_guard<std::mutex> ql(ctx->qmu);
            auto &l = ctx->live;
            l.erase(std::remove(l.begin(), l.end(), s), l.end());
        } catch (...) {
        }
        delete s;
        throw;
    }
    return s;
    BRX_GUARD_END(nullptr, nullptr)
}

extern "C" brx_stream *brx_stream_new(brx_ctx *ctx, const uint8_t *in, size_t n) { return stream_new_impl(ctx, in, n, false); }

extern "C" brx_stream *brx_stream_new_bounded(brx_ctx *ctx, const uint8_t *in, size_t n) { return stream_new_impl(ctx, in, n, true); }

extern "C" brx_stream *brx_stream_new_reader(brx_ctx *ctx, brx_read_fn read, void *user) {
    if (!read) {
        fail(BRX_ERR_INVALID_ARGUMENT, "brx_stream_new_reader: read is NULL");
        return nullptr;
    }
    brx_stream *s = stream_new_impl(ctx, nullptr, 0, true);
    if (s) {
        s->read_fn = read;
        s->read_user = user;
    }
    return s;
}

// Decode every pending stream of the context in one batch; streams whose guessed capacity was too small go into
// the next round with the size the kernel asked for (at least x4), up to the 4 GiB - 256 B per-stream limit.
static void decode_pending_locked(brx_ctx *c, brx_stream *self) {
    std::vector<brx_stream *> todo;
    {
        std::lock_guard<std::mutex> ql(c->qmu); // (under the context's lock: brx_stream_free takes both, so nothing in `todo` goes away)
        todo.swap(c->pending);
        c->facade_prev_n = todo.size();
    }
    std::vector<brx_stream *> all(todo); // what this batch still owns: an entry goes to nullptr the moment its stream is flagged decoded
                                         // (its owner may free it at once)
    try {
    std::vector<size_t> cap(todo.size()), ai(todo.size());
    std::iota(ai.begin(), ai.end(), (size_t)0);
    for (size_t i = 0; i < todo.size(); i++) cap[i] = todo[i]->in.size() * 8 + 65536;
    while (!todo.empty()) {
        const uint32_t n = (uint32_t)std::min<size_t>(todo.size(), 1u << 20);
        std::vector<uint64_t> in_off(n + 1, 0), out_off(n + 1, 0), out_len(n, 0);
        std::vector<int32_t> st(n, 0);
        for (uint32_t i = 0; i < n; i++) {
            in_off[i + 1] = in_off[i] + todo[i]->in.size();
            out_off[i + 1] = out_off[i] + ((cap[i] + 15) & ~(size_t)15);
        }
        // Staging: the context's pinned arenas while the batch fits BRX_FACADE_PINNED_MAX (the kernel reads the compressed bytes and
        // writes the decoded ones in place over PCIe: nothing of the slots' unused capacity is copied, no fresh pages are touched);
        // beyond that, plain host memory for this one batch.
        const size_t in_need = (size_t)in_off[n] + 16, out_need = (size_t)out_off[n] + 16;
        { // the owners of the batch before this one copy their streams out of the arena themselves: they are done in a moment
            std::unique_lock<std::mutex> ql(c->qmu);
            c->fa_cv.wait(ql, [&] { return c->fa_readers == 0; });
        }
        std::unique_ptr<uint8_t[]> in_own, out_own;
        uint8_t *in_p = nullptr, *out_p = nullptr;
        if (in_need + out_need <= BRX_FACADE_PINNED_MAX && hipSetDevice(c->device) == hipSuccess) {
            auto grow_pinned = [](uint8_t **p, size_t *cap, size_t need) {
                if (*p && need <= *cap) return true;
                if (*p) (void)hipHostFree(*p);
                *p = nullptr;
                *cap = 0;
                const size_t want = need + need / 4 + (1u << 20);
                *p = (uint8_t *)brx_host_alloc(want);
                if (*p) *cap = want;
                return *p != nullptr;
            };
            if (grow_pinned(&c->fa_in, &c->fa_in_cap, in_need) && grow_pinned(&c->fa_out, &c->fa_out_cap, out_need)) {
                in_p = c->fa_in;
                out_p = c->fa_out;
            }
        }
        if (!in_p) {
            in_own.reset(new uint8_t[in_need]);
            out_own.reset(new uint8_t[out_need]);
            in_p = in_own.get();
            out_p = out_own.get();
        }
        for (uint32_t i = 0; i < n; i++)
            if (!todo[i]->in.empty()) memcpy(in_p + in_off[i], todo[i]->in.data(), todo[i]->in.size());
        int rc = decode_batch_locked(c, in_p, in_off.data(), n, out_p, out_off.data(), out_len.data(), st.data(), nullptr);
        c->facade_batches++;
        c->facade_streams += n;
        std::vector<brx_stream *> again;
        std::vector<size_t> again_cap, again_ai;
        std::vector<uint32_t> ready; // decoded (or failed with a stream status): their bytes are in the staging
        for (uint32_t i = 0; i < n; i++) {
            brx_stream *s = todo[i];
            if (rc != BRX_SUCCESS) {
                s->lib_rc = rc;
                all[ai[i]] = nullptr;
                s->decoded = true;
                continue;
            }
            if (st[i] == BRX_OUTPUT_TOO_SMALL && cap[i] < BRX_STREAM_LIMIT) {
                size_t want = std::max<size_t>(cap[i] * 4, (size_t)out_len[i]);
                again.push_back(s);
                again_cap.push_back((size_t)std::min<uint64_t>(want, BRX_STREAM_LIMIT));
                again_ai.push_back(ai[i]);
                continue;
            }
            if (st[i] == BRX_OUTPUT_TOO_SMALL) { // the stream expands past the per-stream limit: a definite error
                s->lib_rc = BRX_ERR_OUT_OF_MEMORY;
                all[ai[i]] = nullptr;
                s->decoded = true;
                continue;
            }
            s->status = st[i];
            ready.push_back(i);
        }
        // Who copies a stream out of the staging: its owner, when the owner is blocked in brx_stream_read right now (N owners copy at
        // once; the arena is the context's and nothing else wants it before the next batch, which waits for them) -- else this thread.
        std::vector<uint8_t> by_owner(n, 0);
        if (out_p == c->fa_out && again.empty() && todo.size() == n) {
            std::lock_guard<std::mutex> ql(c->qmu);
            for (uint32_t i : ready) {
                brx_stream *s = todo[i];
                if (!s->waiting && s != self) continue;
                s->view = out_p + out_off[i];
                s->view_len = (size_t)std::min<uint64_t>(out_len[i], cap[i]);
                c->fa_readers++;
                by_owner[i] = 1;
            }
        }
        for (uint32_t i : ready) {
            brx_stream *s = todo[i];
            if (!by_owner[i]) {
                const size_t produced = (size_t)std::min<uint64_t>(out_len[i], cap[i]);
                s->out.assign(out_p + out_off[i], out_p + out_off[i] + produced);
            }
            std::vector<uint8_t>().swap(s->in);
            all[ai[i]] = nullptr;
            s->decoded = true; // (last: from here on the stream is its owner's alone -- it may be read and freed at once)
        }
        for (size_t i = n; i < todo.size(); i++) { // (more than 2^20 pending streams: next round)
            again.push_back(todo[i]);
            again_cap.push_back(cap[i]);
            again_ai.push_back(ai[i]);
        }
        todo.swap(again);
        cap.swap(again_cap);
        ai.swap(again_ai);
    }
    } catch (...) { // (a host allocation failed: no owner may be left waiting for a stream this batch took)
        for (brx_stream *s : all)
            if (s && !s->decoded) {
                s->lib_rc = BRX_ERR_OUT_OF_MEMORY;
                s->decoded = true;
            }
        throw;
    }
}

extern "C" int64_t brx_stream_read(brx_stream *s, uint8_t *buf, size_t len) {
    BRX_GUARD_BEGIN
    if (!s) return -(int64_t)BRX_UNEXPECTED_EOF;
    if (s->bounded) {
        brx_ctx *c = s->ctx;
        if (!c) return -(int64_t)1000 + BRX_ERR_INVALID_ARGUMENT;
        std::unique_lock<std::mutex> lk(c->mu);
        for (;;) {
            if (s->lib_rc != BRX_SUCCESS) return -(int64_t)1000 + s->lib_rc;
            if (s->delivered < s->pos) {
                const size_t k = (size_t)std::min<uint64_t>(len, s->pos - s->delivered);
                if (k == 0) return 0;
                if (hipSetDevice(c->device) != hipSuccess ||
                    hipMemcpy(buf, s->d_buf + (s->delivered - s->shift), k, hipMemcpyDeviceToHost) != hipSuccess) {
                    s->lib_rc = fail(BRX_ERR_HIP, "brx_stream_read: device to host copy failed");
                    continue;
                }
                s->delivered += k;
                return (int64_t)k;
            }
            if (s->finished) {
                bounded_release(s);
                return s->status == BRX_OK ? 0 : -(int64_t)s->status; // the bytes before an error were served first
            }
            if (!s->d_buf) {
                int rc = bounded_init(s);
                if (rc) { bounded_release(s); s->lib_rc = rc; continue; }
            }
            int rc = bounded_step(s, lk);
            if (rc == BRX_ERR_OUT_OF_MEMORY && !s->finished && s->read_fn != nullptr) {
                // (over a reader the compressed bytes behind the window are gone: no second way)
                bounded_release(s);
                s->lib_rc = fail(BRX_ERR_OUT_OF_MEMORY, "brx_stream_read: one item of the stream needs more device memory than the bounded reader may take (or an allocation failed); decode this stream from memory (brx_stream_new)");
                continue;
            }
            if (rc == BRX_ERR_OUT_OF_MEMORY && !s->finished) {
                // a single command (a copy or an uncompressed meta-block) larger than the slack: decode the whole stream
                // the unbounded way and go on serving from where this reader stands
                bounded_release(s);
                s->bounded = false;
                s->served = (size_t)s->delivered;
                {
                    std::lock_guard<std::mutex> ql(c->qmu);
                    c->pending.push_back(s);
                }
                decode_pending_locked(c, s);
                c->qcv.notify_all(); // (streams of other threads went out with it)
                break;
            }
            if (rc) { bounded_release(s); s->lib_rc = rc; }
        }
    }
    if (!s->decoded.load(std::memory_order_acquire)) {
        // One reader leads: it decodes everything queued on the context -- its own stream and whatever other threads have made since the
        // last batch went out -- while the others wait for THEIR stream on the queue's condition variable (not on the context's lock:
        // a reader whose stream is done must not sit out the next batch).  Streams made while a batch runs queue up behind it and go
        // out together as the next one: host threads that each make a Decompressor and read it coalesce into batches by themselves
        // (512 threads on alice29: 24 MB/s -> GB/s, tests/cpp/stream_threads.cpp).
        brx_ctx *c = s->ctx;
        if (!c) return -(int64_t)1000 + BRX_ERR_INVALID_ARGUMENT;
        std::unique_lock<std::mutex> ql(c->qmu);
        auto unlist = [&] {
            if (s->waiting) {
                auto &q = c->waitq;
                q.erase(std::remove(q.begin(), q.end(), s), q.end());
                s->waiting = false;
            }
        };
        auto batch_over = [&] { // wake the owners of what is decoded, and the longest-waiting other reader: it leads the next batch
            c->facade_busy = false;
            bool leader = false;
            auto &q = c->waitq;
            size_t keep = 0;
            for (size_t i = 0; i < q.size(); i++) {
                brx_stream *w = q[i];
                if (w->decoded.load(std::memory_order_acquire)) {
                    w->waiting = false;
                    w->cv.notify_one();
                    continue;
                }
                if (!leader) {
                    leader = true;
                    w->cv.notify_one();
                }
                q[keep++] = w;
            }
            q.resize(keep);
            c->qcv.notify_all(); // (brx_stream_free of a stream inside the batch)
        };
        while (!s->decoded.load(std::memory_order_acquire)) {
            if (c->facade_busy) {
                if (!s->waiting) {
                    s->waiting = true;
                    c->waitq.push_back(s);
                }
                s->cv.wait(ql);
                continue;
            }
            unlist();
            c->facade_busy = true;
            // Where threads work in a closed loop (make, read, free, again) the owners of the batch that has just ended are back
            // within a few hundred microseconds; a batch costs about the same whether it carries half of the threads or all of
            // them, so the leader gives them that moment: it waits while streams keep arriving (quiet for 120 us = go), 1.5 ms at
            // most -- and only when there IS company (the last batch had more than one stream, or more than one is queued: a lone
            // Decompressor never waits) but not a crowd (from ~100 streams a batch on, the host's copies begin to weigh as much as the kernel
            // and threads that outnumber the cores do not come back in time: 512 threads lost a quarter with the wait).
            const size_t company = std::max(c->facade_prev_n, c->pending.size());
            if (company > 1 && company < BRX_FACADE_LINGER_BELOW) {
                const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(BRX_FACADE_LINGER_MAX_US);
                size_t seen = c->pending.size();
                for (;;) {
                    c->new_cv.wait_for(ql, std::chrono::microseconds(BRX_FACADE_LINGER_QUIET_US));
                    if (c->pending.size() == seen || std::chrono::steady_clock::now() >= t_end) break;
                    seen = c->pending.size();
                }
            }
            ql.unlock();
            try {
                std::lock_guard<std::mutex> lk(c->mu);
                decode_pending_locked(c, s);
            } catch (...) {
                ql.lock();
                batch_over();
                throw;
            }
            ql.lock();
            batch_over();
            if (!s->decoded.load(std::memory_order_acquire)) { // (cannot happen: a stream that is not decoded is pending)
                s->lib_rc = BRX_ERR_INVALID_ARGUMENT;
                s->decoded.store(true, std::memory_order_release);
            }
        }
        unlist();
    }
    if (s->view) { // this thread was waiting when its batch ended: the stream's bytes are still in the context's staging
        brx_ctx *c = s->ctx;
        bool ok = true;
        try {
            s->out.assign(s->view, s->view + s->view_len);
        } catch (...) {
            ok = false;
        }
        s->view = nullptr;
        if (c) {
            std::lock_guard<std::mutex> ql(c->qmu);
            if (--c->fa_readers == 0) c->fa_cv.notify_all();
        }
        if (!ok) s->lib_rc = BRX_ERR_OUT_OF_MEMORY;
    }
    if (s->lib_rc != BRX_SUCCESS) {
        if (s->lib_rc == BRX_ERR_OUT_OF_MEMORY) fail(s->lib_rc, "stream expands past the 4 GiB - 256 B per-stream output limit (or allocation failed)");
        return -(int64_t)1000 + s->lib_rc; // library failure, not a stream status
    }
    size_t n = s->out.size() - s->served;
    if (n == 0 && s->status != BRX_OK) return -(int64_t)s->status; // after the bytes produced before the error
    if (n > len) n = len;
    if (n) memcpy(buf, s->out.data() + s->served, n);
    s->served += n;
    return (int64_t)n;
    BRX_GUARD_END(-(int64_t)1000 + BRX_ERR_OUT_OF_MEMORY, -(int64_t)1000 + BRX_ERR_HIP)
}

extern "C" void brx_stream_free(brx_stream *s) {
    if (!s) return;
    bounded_release(s);
    try {
        if (brx_ctx *c = s->ctx) {
            // Only the queue's lock: freeing a stream that has been read does not wait for the batch that is running (its owner's
            // next stream would miss that batch's successor).  A stream that is neither decoded nor pending IS in the running batch:
            // that one is waited for.
            std::unique_lock<std::mutex> ql(c->qmu);
            auto &p = c->pending;
            const auto at = std::find(p.begin(), p.end(), s);
            if (at != p.end()) p.erase(at);
            else if (!s->bounded)
                while (!s->decoded.load(std::memory_order_acquire)) c->qcv.wait(ql);
            auto &l = c->live;
            l.erase(std::remove(l.begin(), l.end(), s), l.end());
        }
    } catch (...) {
    }
    delete s;
}
