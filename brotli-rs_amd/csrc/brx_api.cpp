// brx_api.cpp -- host side of the C ABI declared in include/brx.h.
//
// Owns the per-GPU context (device copies of the constant tables, the per-wave spill arena, a stream)
// and implements brx_decode_batch + the Read-shaped stream facade on top of the gfx950 kernel in
// brx_kernels.hip.  There is deliberately NO CPU decode path in this library: without a HIP device
// brx_ctx_create fails with BRX_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <numeric>
#include <new>
#include <string>
#include <vector>

#include "../../include/brx.h"
#include "_gen/brx_tables_gen.h" // BRX_DICT, BRX_CONTEXT_LUT, BRX_TRANSFORMS  (tools/bin2h.py from tables/*.bin)
#include "brx_device.h"

static thread_local std::string g_err;

static int fail(int code, const char *what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return code;
}

#define HIP_TRY(call)                                           \
    do {                                                        \
        hipError_t e_ = (call);                                 \
        if (e_ != hipSuccess) return fail(BRX_ERR_HIP, #call, e_); \
    } while (0)

struct brx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    uint8_t *d_dict = nullptr;
    uint8_t *d_lut = nullptr;
    BrxTransform *d_xforms = nullptr;
    uint32_t *d_iac = nullptr;
    uint32_t *d_counter = nullptr;
    uint32_t *d_scratch = nullptr;
    unsigned max_grid = 0; // resident waves we size the spill arena for
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool have_timing = false;
    // host-mode staging buffers (grown on demand)
    uint8_t *st_in = nullptr, *st_out = nullptr;
    uint64_t *st_meta = nullptr; // in_off | out_off | out_len, then status
    size_t st_in_cap = 0, st_out_cap = 0, st_meta_cap = 0;
};

extern "C" const char *brx_last_error(void) { return g_err.c_str(); }

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

extern "C" int brx_ctx_create(brx_ctx **out, int device) {
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
    c->device = device;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    // 16 single-wave workgroups per CU: 4 per SIMD, bounded by the ~10 KiB of LDS each one declares.
    c->max_grid = (unsigned)prop.multiProcessorCount * 16u;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(&c->d_dict, sizeof BRX_DICT));
    HIP_TRY(hipMalloc(&c->d_lut, sizeof BRX_CONTEXT_LUT));
    HIP_TRY(hipMalloc(&c->d_xforms, 121 * sizeof(BrxTransform)));
    HIP_TRY(hipMalloc(&c->d_counter, 256));
    HIP_TRY(hipMalloc(&c->d_scratch, (size_t)c->max_grid * BRX_SCRATCH_WORDS * 4u));
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
        std::vector<uint32_t> t(704 * 4 + 64, 0u);
        for (unsigned sym = 0; sym < 704; sym++) {
            unsigned cell = sym >> 6, ic = cell_ins[cell] + ((sym >> 3) & 7u), cc = cell_cpy[cell] + (sym & 7u);
            t[4 * sym] = ins_base[ic];
            t[4 * sym + 1] = ins_extra[ic];
            t[4 * sym + 2] = cpy_base[cc];
            t[4 * sym + 3] = cpy_extra[cc];
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
    *out = c;
    return BRX_SUCCESS;
}

extern "C" void brx_ctx_destroy(brx_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->d_dict);
    (void)hipFree(c->d_lut);
    (void)hipFree(c->d_xforms);
    (void)hipFree(c->d_iac);
    (void)hipFree(c->d_counter);
    (void)hipFree(c->d_scratch);
    (void)hipFree(c->st_in);
    (void)hipFree(c->st_out);
    (void)hipFree(c->st_meta);
    for (auto &ev : c->ev)
        if (ev) (void)hipEventDestroy(ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
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

static int launch(brx_ctx *c, hipStream_t st, bool timing, const uint8_t *d_in, const uint64_t *d_in_off, uint32_t n,
                  uint8_t *d_out, const uint64_t *d_out_off, uint64_t *d_out_len, int32_t *d_status,
                  const uint32_t *d_order = nullptr) {
    BrxKernelArgs a;
    a.order = d_order;
    a.in = d_in;
    a.in_off = d_in_off;
    a.out = d_out;
    a.out_off = d_out_off;
    a.out_len = d_out_len;
    a.status = d_status;
    a.n = n;
    {
        const char *e = getenv("BRX_DEBUG_STOP");
        a.debug_stop = e ? (uint32_t)atoi(e) : 0u;
    }
    a.work_counter = c->d_counter;
    a.scratch = c->d_scratch;
    a.debug = nullptr;
    unsigned long long *dbg = nullptr;
    if (getenv("BRX_DEBUG_STATS")) {
        if (hipMalloc(&dbg, (size_t)n * 80) == hipSuccess) { (void)hipMemset(dbg, 0, (size_t)n * 80); a.debug = dbg; }
    }
    a.t.dict = c->d_dict;
    a.t.context_lut = c->d_lut;
    a.t.xforms = c->d_xforms;
    a.t.iac = c->d_iac;
    unsigned grid = n < c->max_grid ? n : c->max_grid;
    HIP_TRY(hipMemsetAsync(c->d_counter, 0, 4, st));
    if (timing) HIP_TRY(hipEventRecord(c->ev[2], st));
    brx_launch_decode(a, grid, st);
    HIP_TRY(hipGetLastError());
    if (timing) HIP_TRY(hipEventRecord(c->ev[3], st));
    if (dbg) {
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h((size_t)n * 10);
        (void)hipMemcpy(h.data(), dbg, (size_t)n * 80, hipMemcpyDeviceToHost);
        static const char *nm[10] = {"hdr", "iac", "lit", "dist", "copy", "ncmd", "nlit", "fastmb", "total", "scr_top"};
        for (uint32_t i = 0; i < n && i < (getenv("BRX_DEBUG_STATS_ALL") ? n : 2u); i++) {
            fprintf(stderr, "[brx stats] stream %u:", i);
            for (int q = 0; q < 10; q++) fprintf(stderr, " %s=%llu", nm[q], h[(size_t)i * 10 + q]);
            fprintf(stderr, "\n[brx stats] words:");
            for (int q = 0; q < 8; q++) fprintf(stderr, " %u %u", (unsigned)h[(size_t)i * 10 + q], (unsigned)(h[(size_t)i * 10 + q] >> 32));
            fprintf(stderr, "\n");
        }
        (void)hipFree(dbg);
    }
    return BRX_SUCCESS;
}

extern "C" int brx_decode_batch(brx_ctx *c, const uint8_t *in, const uint64_t *in_off, uint32_t n, uint8_t *out,
                                const uint64_t *out_off, uint64_t *out_len, int32_t *status, const brx_opts *opts) {
    if (!c) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_decode_batch: ctx is NULL");
    if (n == 0) return BRX_SUCCESS;
    if (!in_off || !out_off || !out_len || !status) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_decode_batch: NULL table");
    const uint32_t flags = opts ? opts->flags : 0u;
    const bool timing = (flags & BRX_OPT_TIMING) != 0;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = (opts && opts->hip_stream) ? (hipStream_t)opts->hip_stream : c->stream;
    c->have_timing = false;

    if (flags & BRX_MEM_DEVICE) {
        if (timing) HIP_TRY(hipEventRecord(c->ev[0], st));
        int rc = launch(c, st, timing, in, in_off, n, out, out_off, out_len, status);
        if (rc) return rc;
        if (timing) HIP_TRY(hipEventRecord(c->ev[1], st));
        if (!(opts && opts->hip_stream)) HIP_TRY(hipStreamSynchronize(st));
        c->have_timing = timing;
        return BRX_SUCCESS;
    }

    // ---- host pointers: stage through HBM (H2D, decode, D2H); PCIe time is NOT part of any reported rate
    for (uint32_t i = 0; i < n; i++)
        if (in_off[i + 1] < in_off[i] || out_off[i + 1] < out_off[i])
            return fail(BRX_ERR_INVALID_ARGUMENT, "brx_decode_batch: offsets must be non-decreasing");
    const uint64_t in_lo = in_off[0], in_hi = in_off[n], out_lo = out_off[0], out_hi = out_off[n];
    const size_t in_bytes = (size_t)(in_hi - in_lo), out_bytes = (size_t)(out_hi - out_lo);
    if ((in_bytes && !in) || (out_bytes && !out)) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_decode_batch: NULL data");
    int rc;
    if ((rc = grow(&c->st_in, &c->st_in_cap, in_bytes + 16))) return rc;
    if ((rc = grow(&c->st_out, &c->st_out_cap, out_bytes + 16))) return rc;
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
    // Work-queue order: longest compressed stream first.  The streams of a batch are ragged; a
    // stream is a serial job on one wavefront, so the batch finishes when its longest job does -- start those first.
    uint32_t *d_order = (uint32_t *)(d_status + n);
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        return in_off[x + 1] - in_off[x] > in_off[y + 1] - in_off[y];
    });
    HIP_TRY(hipMemcpyAsync(d_order, order.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    if (in_bytes) HIP_TRY(hipMemcpyAsync(c->st_in, in + in_lo, in_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_in_off, hmeta.data(), hmeta.size() * 8, hipMemcpyHostToDevice, st));
    if (timing) HIP_TRY(hipEventRecord(c->ev[0], st));
    rc = launch(c, st, timing, c->st_in, d_in_off, n, c->st_out, d_out_off, d_out_len, d_status,
                getenv("BRX_NO_ORDER") ? nullptr : d_order);
    if (rc) return rc;
    if (timing) HIP_TRY(hipEventRecord(c->ev[1], st));
    HIP_TRY(hipMemcpyAsync(out_len, d_out_len, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(status, d_status, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    if (out_bytes) HIP_TRY(hipMemcpyAsync(out + out_lo, c->st_out, out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    c->have_timing = timing;
    return BRX_SUCCESS;
}

extern "C" double brx_last_timing(brx_ctx *c, int which) {
    if (!c || !c->have_timing) return -1.0;
    float ms = 0.f;
    hipError_t e = which == 0 ? hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) : hipEventElapsedTime(&ms, c->ev[2], c->ev[3]);
    return e == hipSuccess ? (double)ms : -1.0;
}

extern "C" int brx_synchronize(brx_ctx *c, void *hip_stream) {
    if (!c) return fail(BRX_ERR_INVALID_ARGUMENT, "brx_synchronize: ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(hip_stream ? (hipStream_t)hip_stream : c->stream));
    return BRX_SUCCESS;
}

// ---- Read-shaped facade: one object = one stream (reference Decompressor<R>, src/lib.rs:377-410, 2173-2193)
struct brx_stream {
    brx_ctx *ctx;
    std::vector<uint8_t> in;
    std::vector<uint8_t> out;
    size_t served = 0;
    bool decoded = false;
    int32_t status = 0;
};

extern "C" brx_stream *brx_stream_new(brx_ctx *ctx, const uint8_t *in, size_t n) {
    if (!ctx || (n && !in)) {
        fail(BRX_ERR_INVALID_ARGUMENT, "brx_stream_new: bad argument");
        return nullptr;
    }
    brx_stream *s = new (std::nothrow) brx_stream();
    if (!s) return nullptr;
    s->ctx = ctx;
    s->in.assign(in, in + n);
    return s;
}

extern "C" int64_t brx_stream_read(brx_stream *s, uint8_t *buf, size_t len) {
    if (!s) return -(int64_t)BRX_UNEXPECTED_EOF;
    if (!s->decoded) {
        // Brotli carries no total length: start from a guess and grow on BRX_OUTPUT_TOO_SMALL.
        size_t cap = s->in.size() * 8 + 65536;
        for (;;) {
            s->out.resize(cap);
            uint64_t in_off[2] = {0, s->in.size()}, out_off[2] = {0, cap}, out_len = 0;
            int32_t st = 0;
            int rc = brx_decode_batch(s->ctx, s->in.data(), in_off, 1, s->out.data(), out_off, &out_len, &st, nullptr);
            if (rc != BRX_SUCCESS) return -(int64_t)1000 + rc; // library failure, not a stream status
            if (st == BRX_OUTPUT_TOO_SMALL) {
                cap = cap * 4 > out_len ? cap * 4 : (size_t)out_len;
                continue;
            }
            s->status = st;
            s->out.resize(st == BRX_OK ? (size_t)out_len : 0); // no partial output on error
            break;
        }
        s->decoded = true;
    }
    if (s->status != BRX_OK) return -(int64_t)s->status;
    size_t n = s->out.size() - s->served;
    if (n > len) n = len;
    if (n) memcpy(buf, s->out.data() + s->served, n);
    s->served += n;
    return (int64_t)n;
}

extern "C" void brx_stream_free(brx_stream *s) { delete s; }
