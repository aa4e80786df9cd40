// brx_node.cpp -- the GPUs of one machine behind one call: brx_node_* of include/brx.h (round 6).
//
// Streams are independent (reference src/lib.rs:378-394: a Decompressor owns all of its state), so a batch is DEALT over the GPUs and
// the shards are decoded at the same time by the ordinary one-GPU entry point, brx_decode_batch, each on its own context from its own
// host thread.  Nothing here touches a stream's bytes except to move them: with host pointers every GPU reads and writes the caller's
// buffers itself (no exchange at all); with device pointers the root's GPU holds everything and the others get their shard over xGMI
// and send their results back -- one grouped point-to-point exchange each way (RCCL send / recv, or peer copies), never a collective
// inside the decode.  Python's shard.py (torch.distributed, one process per GPU) does the same thing for the driver's bench; this is
// the form a Rust host binds.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types only: librccl is dlopen'ed when the first exchange needs it (a one-GPU caller never loads it)

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "brx_internal.h"

namespace {

// ---- one host thread per rank ---------------------------------------------------------------------------------------------------
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has = false, quit = false, done = true;
    int rc = 0;
    std::string err;

    void run(int device) {
        (void)hipSetDevice(device);
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return has || quit; });
            if (quit) return;
            std::function<int()> j;
            j.swap(job);
            has = false;
            lk.unlock();
            int r;
            std::string e;
            try {
                r = j();
                if (r != BRX_SUCCESS) e = brx_last_error(); // (thread-local: carried over to the caller's thread)
            } catch (const std::bad_alloc &) {
                r = BRX_ERR_OUT_OF_MEMORY;
                e = "host allocation failed";
            } catch (...) {
                r = BRX_ERR_HIP;
                e = "unexpected C++ exception inside libbrx";
            }
            lk.lock();
            rc = r;
            err.swap(e);
            done = true;
            cv.notify_all();
        }
    }
    void post(std::function<int()> j) {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(j);
        has = true;
        done = false;
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        return rc;
    }
};

// ---- what a rank keeps between calls ----------------------------------------------------------------------------------------------
struct Buf { // a device (or pinned host) buffer grown on demand
    uint8_t *p = nullptr;
    size_t cap = 0;
};

struct Rank {
    int device = 0;
    brx_ctx *ctx = nullptr;
    Worker *w = nullptr;
    hipStream_t s = nullptr;           // this rank's exchange / decode stream (device paths)
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    Buf d_in, d_out, d_pack, d_tab;    // shard input, output slots, compacted output, tables (device memory of THIS rank)
    Buf h_in, h_out;                   // BRX_NODE_DEAL_SNAKE under BRX_MEM_HOST: the shard packed in pinned host memory
    // of the most recent call
    uint32_t streams = 0;
    uint64_t in_bytes = 0, packed = 0;
    double kernel_ms = -1.0;
};

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::vector<ncclComm_t> comm; // one per rank of the node
    bool tried = false, ok = false;
};

} // namespace

struct brx_node {
    std::vector<Rank> r;
    std::mutex mu; // one batch at a time
    int transport = 0;
    int64_t min_streams = 0;
    bool exchange_root = false;
    bool distinct = true; // every rank has a GPU of its own
    Rccl rccl;
    Buf root_gather, root_tab, root_pack; // on the root's GPU (freed and re-made when the root changes)
    int root_of_bufs = -1;
    // of the most recent call
    int used = 0;
    double wall_ms = -1.0, scatter_ms = 0.0;
    bool used_rccl = false;
};

namespace {

#define NODE_HIP(call)                                                \
    do {                                                              \
        hipError_t e_ = (call);                                       \
        if (e_ != hipSuccess) {                                       \
            char b_[256];                                             \
            snprintf(b_, sizeof b_, "%s: %s", #call, hipGetErrorString(e_)); \
            return brx_fail(BRX_ERR_HIP, b_);                         \
        }                                                             \
    } while (0)

int grow_dev(Buf &b, size_t need) { // (the current device is the buffer's)
    if (b.p && need <= b.cap) return BRX_SUCCESS;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    const size_t want = need + need / 4 + 4096;
    if (hipMalloc(&b.p, want) != hipSuccess) {
        b.p = nullptr;
        return brx_fail(BRX_ERR_OUT_OF_MEMORY, "brx_node: device allocation failed");
    }
    b.cap = want;
    return BRX_SUCCESS;
}

int grow_pinned(Buf &b, size_t need) {
    if (b.p && need <= b.cap) return BRX_SUCCESS;
    if (b.p) brx_host_free(b.p);
    b.p = nullptr;
    b.cap = 0;
    const size_t want = need + need / 4 + 4096;
    b.p = (uint8_t *)brx_host_alloc(want);
    if (!b.p) return BRX_ERR_OUT_OF_MEMORY;
    b.cap = want;
    return BRX_SUCCESS;
}

// librccl, when an exchange wants it.  One communicator per rank, all made by this process (ncclCommInitAll): the grouped
// ncclSend / ncclRecv pairs below are issued by ONE thread for all GPUs, the single-process form of the pattern shard.py runs with
// one process per GPU.
int rccl_ready(brx_node *nd) {
    Rccl &q = nd->rccl;
    if (q.tried) return q.ok ? BRX_SUCCESS : brx_fail(BRX_ERR_HIP, "brx_node: RCCL is not available (see the first failure)");
    q.tried = true;
    for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
        q.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (q.lib) break;
    }
    if (!q.lib) return brx_fail(BRX_ERR_HIP, "brx_node: librccl.so could not be loaded");
    q.CommInitAll = (decltype(q.CommInitAll))dlsym(q.lib, "ncclCommInitAll");
    q.CommDestroy = (decltype(q.CommDestroy))dlsym(q.lib, "ncclCommDestroy");
    q.GroupStart = (decltype(q.GroupStart))dlsym(q.lib, "ncclGroupStart");
    q.GroupEnd = (decltype(q.GroupEnd))dlsym(q.lib, "ncclGroupEnd");
    q.Send = (decltype(q.Send))dlsym(q.lib, "ncclSend");
    q.Recv = (decltype(q.Recv))dlsym(q.lib, "ncclRecv");
    q.GetErrorString = (decltype(q.GetErrorString))dlsym(q.lib, "ncclGetErrorString");
    if (!q.CommInitAll || !q.CommDestroy || !q.GroupStart || !q.GroupEnd || !q.Send || !q.Recv || !q.GetErrorString)
        return brx_fail(BRX_ERR_HIP, "brx_node: librccl.so lacks a symbol of the point-to-point API");
    std::vector<int> devs;
    for (const Rank &k : nd->r) devs.push_back(k.device);
    q.comm.assign(devs.size(), nullptr);
    const ncclResult_t e = q.CommInitAll(q.comm.data(), (int)devs.size(), devs.data());
    if (e != ncclSuccess) {
        char b[256];
        snprintf(b, sizeof b, "brx_node: ncclCommInitAll: %s", q.GetErrorString(e));
        q.comm.clear();
        return brx_fail(BRX_ERR_HIP, b);
    }
    q.ok = true;
    return BRX_SUCCESS;
}

#define NODE_NCCL(nd, call)                                                                   \
    do {                                                                                      \
        ncclResult_t e_ = (call);                                                             \
        if (e_ != ncclSuccess) {                                                              \
            char b_[256];                                                                     \
            snprintf(b_, sizeof b_, "%s: %s", #call, (nd)->rccl.GetErrorString(e_));          \
            return brx_fail(BRX_ERR_HIP, b_);                                                 \
        }                                                                                     \
    } while (0)

// ---- dealing ---------------------------------------------------------------------------------------------------------------------
// order[k] = the caller's index of the k-th stream of the dealt order; rank r takes order[cut[r] .. cut[r + 1]).
struct Deal {
    std::vector<uint32_t> order; // empty = the identity (contiguous ranges)
    std::vector<uint32_t> cut;
};

Deal deal_streams(const uint64_t *in_off, uint32_t n, int G, uint32_t how) {
    Deal d;
    d.cut.assign((size_t)G + 1, 0);
    for (int r = 0; r <= G; r++) d.cut[r] = (uint32_t)(((uint64_t)r * n) / (uint64_t)G); // [r * n / G, (r + 1) * n / G): shard.py's shard_range
    if (how == BRX_NODE_DEAL_BYTES && G > 1) {
        const uint64_t lo = in_off[0], total = in_off[n] - lo;
        uint32_t i = 0;
        for (int r = 1; r < G; r++) {
            const uint64_t goal = total / (uint64_t)G * (uint64_t)r + total % (uint64_t)G * (uint64_t)r / (uint64_t)G;
            while (i < n && in_off[i] - lo < goal) i++;
            d.cut[r] = i;
        }
    } else if (how == BRX_NODE_DEAL_SNAKE && G > 1) {
        // shard.py's balanced_order: largest first, snake order over the ranks, a rank that has its count is skipped
        std::vector<uint32_t> by_size(n);
        std::iota(by_size.begin(), by_size.end(), 0u);
        std::stable_sort(by_size.begin(), by_size.end(), [&](uint32_t x, uint32_t y) { return in_off[x + 1] - in_off[x] > in_off[y + 1] - in_off[y]; });
        std::vector<std::vector<uint32_t>> bucket((size_t)G);
        int r = 0, step = 1;
        for (uint32_t i : by_size) {
            for (int tries = 0; tries < 2 * G; tries++) {
                if (bucket[r].size() < (size_t)(d.cut[r + 1] - d.cut[r])) break;
                const int nr = r + step;
                if (nr < 0 || nr >= G) step = -step; else r = nr;
            }
            bucket[r].push_back(i);
            const int nr = r + step;
            if (nr < 0 || nr >= G) step = -step; else r = nr;
        }
        d.order.reserve(n);
        for (int k = 0; k < G; k++) d.order.insert(d.order.end(), bucket[k].begin(), bucket[k].end());
    }
    return d;
}

int wait_all(brx_node *nd, const std::vector<int> &ranks) {
    int rc = BRX_SUCCESS;
    std::string err;
    for (int r : ranks) {
        const int k = nd->r[r].w->wait();
        if (k != BRX_SUCCESS && rc == BRX_SUCCESS) {
            rc = k;
            char b[64];
            snprintf(b, sizeof b, "rank %d: ", r);
            err = std::string(b) + nd->r[r].w->err;
        }
    }
    return rc == BRX_SUCCESS ? rc : brx_fail(rc, err.c_str());
}

// ---- host pointers: no exchange ------------------------------------------------------------------------------------------------------
int decode_host(brx_node *nd, const uint8_t *in, const uint64_t *in_off, uint32_t n, uint8_t *out, const uint64_t *out_off,
                uint64_t *out_len, int32_t *status, int G, const Deal &deal, bool timing) {
    std::vector<int> ranks;
    for (int r = 0; r < G; r++) {
        Rank &k = nd->r[r];
        const uint32_t a = deal.cut[r], b = deal.cut[r + 1];
        k.streams = b - a;
        k.in_bytes = 0;
        k.kernel_ms = -1.0;
        if (a == b) continue;
        ranks.push_back(r);
        if (deal.order.empty()) {
            k.in_bytes = in_off[b] - in_off[a];
            k.w->post([=, &k]() -> int { // the caller's buffers as they are: this rank's slice of the tables is a batch of its own
                brx_opts o = {BRX_MEM_HOST | (timing ? BRX_OPT_TIMING : 0u), 0u, nullptr};
                const int rc = brx_decode_batch(k.ctx, in, in_off + a, b - a, out, out_off + a, out_len + a, status + a, &o);
                if (rc == BRX_SUCCESS && timing) k.kernel_ms = brx_last_timing(k.ctx, 1);
                return rc;
            });
            continue;
        }
        const uint32_t *idx = deal.order.data() + a;
        k.w->post([=, &k]() -> int { // scattered streams: packed into this rank's pinned staging, decoded there in place, unpacked
            const uint32_t m = b - a;
            std::vector<uint64_t> io((size_t)m + 1, 0), oo((size_t)m + 1, 0), ol(m, 0);
            std::vector<int32_t> st(m, 0);
            for (uint32_t i = 0; i < m; i++) {
                io[i + 1] = io[i] + (in_off[idx[i] + 1] - in_off[idx[i]]);
                oo[i + 1] = oo[i] + (out_off[idx[i] + 1] - out_off[idx[i]]); // (the caller's capacity EXACTLY: a slot is its stream's capacity)
            }
            k.in_bytes = io[m];
            if (grow_pinned(k.h_in, (size_t)io[m] + 16) || grow_pinned(k.h_out, (size_t)oo[m] + 16))
                return brx_fail(BRX_ERR_OUT_OF_MEMORY, "brx_node: pinned staging allocation failed");
            for (uint32_t i = 0; i < m; i++)
                if (io[i + 1] > io[i]) memcpy(k.h_in.p + io[i], in + in_off[idx[i]], (size_t)(io[i + 1] - io[i]));
            brx_opts o = {BRX_MEM_HOST | (timing ? BRX_OPT_TIMING : 0u), 0u, nullptr};
            const int rc = brx_decode_batch(k.ctx, k.h_in.p, io.data(), m, k.h_out.p, oo.data(), ol.data(), st.data(), &o);
            if (rc != BRX_SUCCESS) return rc;
            if (timing) k.kernel_ms = brx_last_timing(k.ctx, 1);
            for (uint32_t i = 0; i < m; i++) {
                const uint64_t cap = out_off[idx[i] + 1] - out_off[idx[i]];
                const uint64_t have = std::min<uint64_t>(ol[i], cap); // (a failed stream: the bytes in front of the error, as brx_decode_batch leaves them)
                if (have) memcpy(out + out_off[idx[i]], k.h_out.p + oo[i], (size_t)have);
                out_len[idx[i]] = ol[i];
                status[idx[i]] = st[i];
            }
            return BRX_SUCCESS;
        });
    }
    return wait_all(nd, ranks);
}

// ---- device pointers on the root's GPU: scatter, decode, ragged gather ----------------------------------------------------------------
// Tables of a remote shard in Rank::d_tab (8-byte words): in_off[m + 1] | out_off[m + 1] | out_len[m] | pack_off[m] | status[m] (int32)
struct TabAt {
    size_t in_off, out_off, out_len, pack_off, status, bytes;
};
TabAt tab_layout(uint32_t m) {
    TabAt t;
    t.in_off = 0;
    t.out_off = ((size_t)m + 1) * 8;
    t.out_len = t.out_off + ((size_t)m + 1) * 8;
    t.pack_off = t.out_len + (size_t)m * 8;
    t.status = t.pack_off + (size_t)m * 8;
    t.bytes = t.status + (size_t)m * 4 + 16;
    return t;
}

int decode_device(brx_node *nd, const uint8_t *in, const uint64_t *d_in_off, uint32_t n, uint8_t *out, const uint64_t *d_out_off,
                  uint64_t *d_out_len, int32_t *d_status, int G, uint32_t how, int root, hipStream_t user_stream, bool timing) {
    Rank &R = nd->r[root];
    NODE_HIP(hipSetDevice(R.device));
    hipStream_t st = R.s;
    if (user_stream) { // the call's work on the root goes behind the caller's
        NODE_HIP(hipEventRecord(R.ev_out, user_stream));
        NODE_HIP(hipStreamWaitEvent(st, R.ev_out, 0));
    }
    if (nd->root_of_bufs != root) { // (the root-side temporaries live on the root's GPU)
        for (Buf *b : {&nd->root_gather, &nd->root_tab, &nd->root_pack}) {
            if (b->p) {
                if (nd->root_of_bufs >= 0) (void)hipSetDevice(nd->r[nd->root_of_bufs].device);
                (void)hipFree(b->p);
            }
            b->p = nullptr;
            b->cap = 0;
        }
        NODE_HIP(hipSetDevice(R.device));
        nd->root_of_bufs = root;
    }
    // the tables, on the host: the deal needs the sizes
    std::vector<uint64_t> in_off((size_t)n + 1), out_off((size_t)n + 1);
    NODE_HIP(hipMemcpyAsync(in_off.data(), d_in_off, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, st));
    NODE_HIP(hipMemcpyAsync(out_off.data(), d_out_off, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, st));
    NODE_HIP(hipStreamSynchronize(st));
    for (uint32_t i = 0; i < n; i++)
        if (in_off[i + 1] < in_off[i] || out_off[i + 1] < out_off[i])
            return brx_fail(BRX_ERR_INVALID_ARGUMENT, "brx_node_decode_batch: offsets must be non-decreasing");
    const Deal deal = deal_streams(in_off.data(), n, G, how);
    const bool permuted = !deal.order.empty();
    bool use_rccl = false;
    if (G > 1 || nd->exchange_root) {
        if (nd->transport == 2 || (nd->transport == 0 && nd->distinct && G > 1)) {
            if (!nd->distinct) return brx_fail(BRX_ERR_INVALID_ARGUMENT, "brx_node: RCCL needs every rank on a GPU of its own");
            const int rc = rccl_ready(nd);
            if (rc != BRX_SUCCESS && nd->transport == 2) return rc;
            use_rccl = rc == BRX_SUCCESS; // (transport 0: peer copies when librccl is not there)
        }
    }
    nd->used_rccl = use_rccl;
    const auto t_begin = std::chrono::steady_clock::now();
    // A permuted deal: the shards are made contiguous first -- ONE gather over the whole batch on the root, at HBM rate
    const uint8_t *shard_src = in;          // where rank r's bytes start: shard_src + shard_at[r]
    std::vector<uint64_t> dense;            // permuted: exclusive prefix sum of the compressed sizes in dealt order
    if (permuted) {
        dense.assign((size_t)n + 1, 0);
        std::vector<uint64_t> tab(3 * (size_t)n); // src_off | len | dense
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t i = deal.order[k];
            tab[k] = in_off[i];
            tab[(size_t)n + k] = in_off[i + 1] - in_off[i];
            dense[k + 1] = dense[k] + tab[(size_t)n + k];
            tab[2 * (size_t)n + k] = dense[k];
        }
        int rc;
        if ((rc = grow_dev(nd->root_pack, (size_t)dense[n] + 16)) || (rc = grow_dev(nd->root_tab, tab.size() * 8 + 16))) return rc;
        NODE_HIP(hipMemcpyAsync(nd->root_tab.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, st));
        const uint64_t *t = (const uint64_t *)nd->root_tab.p;
        brx_launch_ragged_copy(in, t, t + n, nd->root_pack.p, t + 2 * (size_t)n, t + 2 * (size_t)n, n, dense[n], st);
        NODE_HIP(hipGetLastError());
        NODE_HIP(hipStreamSynchronize(st)); // (`tab` is a local)
        shard_src = nd->root_pack.p;
    }
    auto idx_of = [&](uint32_t k) { return permuted ? deal.order[k] : k; };
    // which ranks work remotely (their shard travels), and what the root decodes in place
    const bool root_in_place = !permuted && !nd->exchange_root;
    std::vector<int> remote, busy;
    for (int r = 0; r < G; r++) {
        Rank &k = nd->r[r];
        k.streams = deal.cut[r + 1] - deal.cut[r];
        k.in_bytes = 0;
        k.packed = 0;
        k.kernel_ms = -1.0;
        if (k.streams == 0) continue;
        busy.push_back(r);
        if (!(r == root && root_in_place)) remote.push_back(r);
    }
    NODE_HIP(hipEventRecord(R.ev_in, st)); // the source bytes are where the shards are read from
    // ---- scatter: every remote rank gets exactly its bytes and its (rebased) tables
    struct Shard {
        uint64_t src_at = 0, bytes = 0, slots = 0;
        std::vector<uint64_t> tab; // host image of in_off' | out_off'
    };
    std::vector<Shard> sh((size_t)G);
    for (int r : remote) {
        Rank &k = nd->r[r];
        const uint32_t a = deal.cut[r], m = k.streams;
        Shard &q = sh[r];
        q.tab.assign(2 * ((size_t)m + 1), 0);
        uint64_t *io = q.tab.data(), *oo = q.tab.data() + m + 1;
        for (uint32_t i = 0; i < m; i++) {
            const uint32_t g = idx_of(a + i);
            io[i + 1] = io[i] + (in_off[g + 1] - in_off[g]);
            oo[i + 1] = oo[i] + (out_off[g + 1] - out_off[g]); // (the caller's capacity EXACTLY: rounded up, a stream that overruns its slot by
                                                               // a few bytes would come back with status 0 and spill into its neighbour's slot at the root)
        }
        q.src_at = permuted ? dense[a] : in_off[a];
        q.bytes = io[m];
        q.slots = oo[m];
        k.in_bytes = q.bytes;
        NODE_HIP(hipSetDevice(k.device));
        const TabAt t = tab_layout(m);
        int rc;
        if ((rc = grow_dev(k.d_in, (size_t)q.bytes + 16)) || (rc = grow_dev(k.d_out, (size_t)q.slots + 16)) || (rc = grow_dev(k.d_tab, t.bytes))) return rc;
        NODE_HIP(hipMemcpyAsync(k.d_tab.p + t.in_off, io, ((size_t)m + 1) * 8, hipMemcpyHostToDevice, k.s));
        NODE_HIP(hipMemcpyAsync(k.d_tab.p + t.out_off, oo, ((size_t)m + 1) * 8, hipMemcpyHostToDevice, k.s));
        if (!use_rccl) {
            NODE_HIP(hipStreamWaitEvent(k.s, R.ev_in, 0));
            if (q.bytes) NODE_HIP(hipMemcpyPeerAsync(k.d_in.p, k.device, shard_src + q.src_at, R.device, (size_t)q.bytes, k.s));
        }
    }
    if (use_rccl && !remote.empty()) { // one group: the root sends every shard, every remote rank receives its own
        Rccl &q = nd->rccl;
        NODE_NCCL(nd, q.GroupStart());
        for (int r : remote) {
            Rank &k = nd->r[r];
            if (sh[r].bytes == 0) continue;
            NODE_NCCL(nd, q.Send(shard_src + sh[r].src_at, (size_t)sh[r].bytes, ncclUint8, r, q.comm[root], st));
            NODE_NCCL(nd, q.Recv(k.d_in.p, (size_t)sh[r].bytes, ncclUint8, root, q.comm[r], k.s));
        }
        NODE_NCCL(nd, q.GroupEnd());
    }
    nd->scatter_ms = 0.0;
    if (timing) { // (only then: the wait serialises what otherwise overlaps)
        for (int r : remote) {
            NODE_HIP(hipSetDevice(nd->r[r].device));
            NODE_HIP(hipStreamSynchronize(nd->r[r].s));
        }
        nd->scatter_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    }
    // ---- decode: every rank on its own thread; a remote rank then learns its sizes and compacts its results
    std::vector<uint64_t> h_len(n, 0);   // (remote streams only) in the CALLER's index
    std::vector<int32_t> h_status(n, 0);
    for (int r : busy) {
        Rank &k = nd->r[r];
        const uint32_t a = deal.cut[r], m = k.streams;
        if (r == root && root_in_place) {
            k.in_bytes = in_off[a + m] - in_off[a];
            k.w->post([=, &k]() -> int {
                brx_opts o = {BRX_MEM_DEVICE | (timing ? BRX_OPT_TIMING : 0u), 0u, st};
                const int rc = brx_decode_batch(k.ctx, in, d_in_off + a, m, out, d_out_off + a, d_out_len + a, d_status + a, &o);
                if (rc == BRX_SUCCESS && timing) {
                    NODE_HIP(hipStreamSynchronize(st));
                    k.kernel_ms = brx_last_timing(k.ctx, 1);
                }
                return rc;
            });
            continue;
        }
        const uint64_t *oo = sh[r].tab.data() + m + 1;
        uint64_t *hl = h_len.data();
        int32_t *hs = h_status.data();
        const uint32_t *order = permuted ? deal.order.data() : nullptr;
        k.w->post([=, &k]() -> int {
            const TabAt t = tab_layout(m);
            uint8_t *tb = k.d_tab.p;
            brx_opts o = {BRX_MEM_DEVICE | (timing ? BRX_OPT_TIMING : 0u), 0u, k.s};
            const int rc = brx_decode_batch(k.ctx, k.d_in.p, (const uint64_t *)(tb + t.in_off), m, k.d_out.p, (const uint64_t *)(tb + t.out_off),
                                            (uint64_t *)(tb + t.out_len), (int32_t *)(tb + t.status), &o);
            if (rc != BRX_SUCCESS) return rc;
            std::vector<uint64_t> len(m), pack((size_t)m + 1, 0);
            std::vector<int32_t> stt(m);
            NODE_HIP(hipMemcpyAsync(len.data(), tb + t.out_len, (size_t)m * 8, hipMemcpyDeviceToHost, k.s));
            NODE_HIP(hipMemcpyAsync(stt.data(), tb + t.status, (size_t)m * 4, hipMemcpyDeviceToHost, k.s));
            NODE_HIP(hipStreamSynchronize(k.s));
            if (timing) k.kernel_ms = brx_last_timing(k.ctx, 1);
            for (uint32_t i = 0; i < m; i++) {
                const uint32_t g = order ? order[a + i] : a + i;
                hl[g] = len[i];
                hs[g] = stt[i];
                len[i] = std::min<uint64_t>(len[i], oo[i + 1] - oo[i]); // what is in the slot (a failed stream: the bytes before the error)
                pack[i + 1] = pack[i] + len[i];
            }
            k.packed = pack[m];
            if (k.packed) { // the results back to back: what travels is exactly the decoded bytes
                int rc2;
                if ((rc2 = grow_dev(k.d_pack, (size_t)k.packed + 16))) return rc2;
                NODE_HIP(hipMemcpyAsync(tb + t.pack_off, pack.data(), (size_t)m * 8, hipMemcpyHostToDevice, k.s));
                NODE_HIP(hipMemcpyAsync(tb + t.out_len, len.data(), (size_t)m * 8, hipMemcpyHostToDevice, k.s));
                brx_launch_ragged_copy(k.d_out.p, (const uint64_t *)(tb + t.out_off), (const uint64_t *)(tb + t.out_len), k.d_pack.p,
                                       (const uint64_t *)(tb + t.pack_off), (const uint64_t *)(tb + t.pack_off), m, k.packed, k.s);
                NODE_HIP(hipGetLastError());
                NODE_HIP(hipStreamSynchronize(k.s)); // (`pack` / `len` are locals)
            }
            return BRX_SUCCESS;
        });
    }
    {
        const int rc = wait_all(nd, busy);
        if (rc != BRX_SUCCESS) return rc;
    }
    // ---- ragged gather: the packed results to the root, then ONE expansion into the caller's slots
    NODE_HIP(hipSetDevice(R.device));
    uint64_t total = 0, streams = 0;
    std::vector<uint64_t> base((size_t)G, 0);
    for (int r : remote) {
        base[r] = total;
        total += nd->r[r].packed;
        streams += nd->r[r].streams;
    }
    std::vector<uint64_t> tab; // src_off (= dense) | len | dst_off, one entry per remote stream; then the results of the remote streams
    if (streams) {
        int rc;
        if ((rc = grow_dev(nd->root_gather, (size_t)total + 16))) return rc;
        tab.assign(3 * (size_t)streams, 0);
        uint64_t at = 0, k = 0;
        for (int r : remote) {
            const uint32_t a = deal.cut[r], m = nd->r[r].streams;
            const uint64_t *oo = sh[r].tab.data() + m + 1;
            for (uint32_t i = 0; i < m; i++, k++) {
                const uint32_t g = idx_of(a + i);
                const uint64_t l = std::min<uint64_t>(h_len[g], oo[i + 1] - oo[i]);
                tab[k] = at;
                tab[streams + k] = l;
                tab[2 * streams + k] = out_off[g];
                at += l;
            }
        }
        if ((rc = grow_dev(nd->root_tab, tab.size() * 8 + 16))) return rc;
        if (!use_rccl) {
            for (int r : remote) {
                Rank &q = nd->r[r];
                if (q.packed == 0) continue;
                NODE_HIP(hipSetDevice(q.device));
                NODE_HIP(hipMemcpyPeerAsync(nd->root_gather.p + base[r], R.device, q.d_pack.p, q.device, (size_t)q.packed, q.s));
                NODE_HIP(hipEventRecord(q.ev_out, q.s));
            }
            NODE_HIP(hipSetDevice(R.device));
            for (int r : remote)
                if (nd->r[r].packed) NODE_HIP(hipStreamWaitEvent(st, nd->r[r].ev_out, 0));
        } else {
            Rccl &q = nd->rccl;
            NODE_NCCL(nd, q.GroupStart());
            for (int r : remote) {
                Rank &k2 = nd->r[r];
                if (k2.packed == 0) continue;
                NODE_NCCL(nd, q.Send(k2.d_pack.p, (size_t)k2.packed, ncclUint8, root, q.comm[r], k2.s));
                NODE_NCCL(nd, q.Recv(nd->root_gather.p + base[r], (size_t)k2.packed, ncclUint8, r, q.comm[root], st));
            }
            NODE_NCCL(nd, q.GroupEnd());
        }
        NODE_HIP(hipMemcpyAsync(nd->root_tab.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, st));
        const uint64_t *t = (const uint64_t *)nd->root_tab.p;
        brx_launch_ragged_copy(nd->root_gather.p, t, t + streams, out, t + 2 * streams, t, (uint32_t)streams, total, st);
        NODE_HIP(hipGetLastError());
        // lengths and status codes of the remote streams into the caller's device tables
        if (!permuted) {
            for (int r : remote) {
                const uint32_t a = deal.cut[r], m = nd->r[r].streams;
                NODE_HIP(hipMemcpyAsync(d_out_len + a, h_len.data() + a, (size_t)m * 8, hipMemcpyHostToDevice, st));
                NODE_HIP(hipMemcpyAsync(d_status + a, h_status.data() + a, (size_t)m * 4, hipMemcpyHostToDevice, st));
            }
        } else { // (every rank was remote: the host has them all)
            NODE_HIP(hipMemcpyAsync(d_out_len, h_len.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
            NODE_HIP(hipMemcpyAsync(d_status, h_status.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
        }
    }
    NODE_HIP(hipStreamSynchronize(st));
    if (use_rccl) // (the sends of the gather sit on the ranks' streams)
        for (int r : remote) {
            NODE_HIP(hipSetDevice(nd->r[r].device));
            NODE_HIP(hipStreamSynchronize(nd->r[r].s));
        }
    NODE_HIP(hipSetDevice(R.device));
    return BRX_SUCCESS;
}

void node_release(brx_node *nd) {
    if (!nd) return;
    if (nd->rccl.ok) {
        for (size_t r = 0; r < nd->rccl.comm.size(); r++)
            if (nd->rccl.comm[r]) {
                (void)hipSetDevice(nd->r[r].device);
                (void)nd->rccl.CommDestroy(nd->rccl.comm[r]);
            }
    }
    // (librccl stays loaded: unloading a library that owns threads is not worth the risk)
    if (nd->root_of_bufs >= 0) {
        (void)hipSetDevice(nd->r[nd->root_of_bufs].device);
        (void)hipFree(nd->root_gather.p);
        (void)hipFree(nd->root_tab.p);
        (void)hipFree(nd->root_pack.p);
    }
    for (Rank &k : nd->r) {
        if (k.w) {
            {
                std::lock_guard<std::mutex> lk(k.w->mu);
                k.w->quit = true;
                k.w->cv.notify_all();
            }
            if (k.w->th.joinable()) k.w->th.join();
            delete k.w;
        }
        (void)hipSetDevice(k.device);
        if (k.s) (void)hipStreamSynchronize(k.s);
        for (Buf *b : {&k.d_in, &k.d_out, &k.d_pack, &k.d_tab}) (void)hipFree(b->p);
        brx_host_free(k.h_in.p);
        brx_host_free(k.h_out.p);
        if (k.ev_in) (void)hipEventDestroy(k.ev_in);
        if (k.ev_out) (void)hipEventDestroy(k.ev_out);
        if (k.s) (void)hipStreamDestroy(k.s);
        if (k.ctx) brx_ctx_destroy(k.ctx);
    }
    delete nd;
}

} // namespace

#define NODE_GUARD_BEGIN try {
#define NODE_GUARD_END                                                            \
    }                                                                             \
    catch (const std::bad_alloc &) {                                              \
        return brx_fail(BRX_ERR_OUT_OF_MEMORY, "host allocation failed");         \
    }                                                                             \
    catch (...) {                                                                 \
        return brx_fail(BRX_ERR_HIP, "unexpected C++ exception inside libbrx");   \
    }

extern "C" int brx_node_create(brx_node **out, const int *devices, int n_devices) {
    NODE_GUARD_BEGIN
    if (!out) return brx_fail(BRX_ERR_INVALID_ARGUMENT, "brx_node_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return brx_fail(BRX_ERR_NO_DEVICE, "no HIP device: libbrx has no CPU fallback");
    std::vector<int> devs;
    if (!devices || n_devices <= 0) {
        for (int d = 0; d < ndev; d++) devs.push_back(d);
    } else {
        if (n_devices > 64) return brx_fail(BRX_ERR_INVALID_ARGUMENT, "brx_node_create: at most 64 ranks");
        devs.assign(devices, devices + n_devices);
    }
    for (int d : devs)
        if (d < 0 || d >= ndev) return brx_fail(BRX_ERR_INVALID_ARGUMENT, "brx_node_create: bad device index");
    brx_node *nd = new brx_node();
    nd->r.resize(devs.size());
    for (size_t r = 0; r < devs.size(); r++) {
        nd->r[r].device = devs[r];
        for (size_t q = 0; q < r; q++)
            if (devs[q] == devs[r]) nd->distinct = false;
    }
    for (size_t r = 0; r < devs.size(); r++) {
        Rank &k = nd->r[r];
        int rc = brx_ctx_create(&k.ctx, k.device);
        if (rc == BRX_SUCCESS && (hipSetDevice(k.device) != hipSuccess || hipStreamCreateWithFlags(&k.s, hipStreamNonBlocking) != hipSuccess ||
                                  hipEventCreateWithFlags(&k.ev_in, hipEventDisableTiming) != hipSuccess ||
                                  hipEventCreateWithFlags(&k.ev_out, hipEventDisableTiming) != hipSuccess))
            rc = brx_fail(BRX_ERR_HIP, "brx_node_create: stream / event creation failed");
        if (rc != BRX_SUCCESS) {
            const std::string keep = brx_last_error();
            node_release(nd);
            return brx_fail(rc, keep.c_str());
        }
        for (size_t q = 0; q < devs.size(); q++) // xGMI peers see each other's memory directly where the platform allows it
            if (devs[q] != k.device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, k.device, devs[q]) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(devs[q], 0);
                (void)hipGetLastError(); // ("already enabled" is not a failure)
            }
        k.w = new Worker();
        k.w->th = std::thread([w = k.w, d = k.device] { w->run(d); });
    }
    *out = nd;
    return BRX_SUCCESS;
    NODE_GUARD_END
}

extern "C" void brx_node_destroy(brx_node *nd) {
    try {
        node_release(nd);
    } catch (...) {
    }
}

extern "C" int brx_node_size(const brx_node *nd) { return nd ? (int)nd->r.size() : 0; }

extern "C" brx_ctx *brx_node_ctx(brx_node *nd, int rank) {
    if (!nd || rank < 0 || rank >= (int)nd->r.size()) {
        brx_fail(BRX_ERR_INVALID_ARGUMENT, "brx_node_ctx: bad rank");
        return nullptr;
    }
    return nd->r[rank].ctx;
}

extern "C" int brx_node_set_option(brx_node *nd, uint32_t option, int64_t value) {
    NODE_GUARD_BEGIN
    if (!nd) return brx_fail(BRX_ERR_INVALID_ARGUMENT, "brx_node_set_option: node is NULL");
    std::lock_guard<std::mutex> lk(nd->mu);
    switch (option) {
    case BRX_NODE_OPTION_TRANSPORT:
        if (value < 0 || value > 2) return brx_fail(BRX_ERR_INVALID_ARGUMENT, "brx_node_set_option: transport is 0, 1 or 2");
        nd->transport = (int)value;
        return BRX_SUCCESS;
    case BRX_NODE_OPTION_MIN_STREAMS:
        if (value < 0) return brx_fail(BRX_ERR_INVALID_ARGUMENT, "brx_node_set_option: min streams >= 0");
        nd->min_streams = value;
        return BRX_SUCCESS;
    case BRX_NODE_OPTION_EXCHANGE_ROOT:
        nd->exchange_root = value != 0;
        return BRX_SUCCESS;
    default:
        for (Rank &k : nd->r) {
            const int rc = brx_ctx_set_option(k.ctx, option, value);
            if (rc != BRX_SUCCESS) return rc;
        }
        return BRX_SUCCESS;
    }
    NODE_GUARD_END
}

extern "C" int brx_node_decode_batch(brx_node *nd, const uint8_t *in, const uint64_t *in_off, uint32_t n, uint8_t *out,
                                     const uint64_t *out_off, uint64_t *out_len, int32_t *status, const brx_node_opts *opts) {
    NODE_GUARD_BEGIN
    if (!nd) return brx_fail(BRX_ERR_INVALID_ARGUMENT, "brx_node_decode_batch: node is NULL");
    if (n == 0) return BRX_SUCCESS;
    if (!in_off || !out_off || !out_len || !status) return brx_fail(BRX_ERR_INVALID_ARGUMENT, "brx_node_decode_batch: NULL table");
    const uint32_t flags = opts ? opts->flags : 0u, how = opts ? opts->deal : BRX_NODE_DEAL_RANGES;
    if (how > BRX_NODE_DEAL_SNAKE) return brx_fail(BRX_ERR_INVALID_ARGUMENT, "brx_node_decode_batch: unknown deal");
    const int size = (int)nd->r.size();
    int G = opts ? opts->use_gpus : 0;
    const int root = opts ? opts->root : 0;
    std::lock_guard<std::mutex> lk(nd->mu);
    if (G == 0) { // one GPU per `min_streams` streams: fewer than a GPU decodes at a time buy next to nothing
        const uint64_t per = nd->min_streams > 0 ? (uint64_t)nd->min_streams : (uint64_t)std::max(1u, brx_ctx_max_grid(nd->r[0].ctx));
        G = (int)std::min<uint64_t>((uint64_t)size, std::max<uint64_t>(1, ((uint64_t)n + per - 1) / per));
        if ((flags & BRX_MEM_DEVICE) && root >= G) G = root + 1;
    }
    if (G < 1 || G > size) return brx_fail(BRX_ERR_INVALID_ARGUMENT, "brx_node_decode_batch: use_gpus exceeds the node");
    if ((flags & BRX_MEM_DEVICE) && (root < 0 || root >= G)) return brx_fail(BRX_ERR_INVALID_ARGUMENT, "brx_node_decode_batch: the root is not one of the ranks used");
    const bool timing = (flags & BRX_OPT_TIMING) != 0;
    nd->used = G;
    nd->used_rccl = false;
    nd->scatter_ms = 0.0;
    for (Rank &k : nd->r) {
        k.streams = 0;
        k.in_bytes = 0;
        k.kernel_ms = -1.0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    int rc;
    if (flags & BRX_MEM_DEVICE) {
        rc = decode_device(nd, in, in_off, n, out, out_off, out_len, status, G, how, root, opts ? (hipStream_t)opts->hip_stream : nullptr, timing);
    } else {
        for (uint32_t i = 0; i < n; i++)
            if (in_off[i + 1] < in_off[i] || out_off[i + 1] < out_off[i])
                return brx_fail(BRX_ERR_INVALID_ARGUMENT, "brx_node_decode_batch: offsets must be non-decreasing");
        if ((in_off[n] > in_off[0] && !in) || (out_off[n] > out_off[0] && !out)) return brx_fail(BRX_ERR_INVALID_ARGUMENT, "brx_node_decode_batch: NULL data");
        const Deal deal = deal_streams(in_off, n, G, how);
        rc = decode_host(nd, in, in_off, n, out, out_off, out_len, status, G, deal, timing);
    }
    nd->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
    NODE_GUARD_END
}

extern "C" int brx_node_deal(const uint64_t *in_off, uint32_t n, int gpus, uint32_t how, uint32_t *order, uint32_t *cut) {
    NODE_GUARD_BEGIN
    if (!in_off || !order || !cut || gpus < 1 || how > BRX_NODE_DEAL_SNAKE) return brx_fail(BRX_ERR_INVALID_ARGUMENT, "brx_node_deal: bad argument");
    for (uint32_t i = 0; i < n; i++)
        if (in_off[i + 1] < in_off[i]) return brx_fail(BRX_ERR_INVALID_ARGUMENT, "brx_node_deal: offsets must be non-decreasing");
    const Deal d = deal_streams(in_off, n, gpus, how);
    for (uint32_t k = 0; k < n; k++) order[k] = d.order.empty() ? k : d.order[k];
    for (int r = 0; r <= gpus; r++) cut[r] = d.cut[r];
    return BRX_SUCCESS;
    NODE_GUARD_END
}

extern "C" double brx_node_last_timing(brx_node *nd, int which, int rank) {
    if (!nd) return -1.0;
    std::lock_guard<std::mutex> lk(nd->mu);
    if (which == 0) return (double)nd->used;
    if (which == 4) return nd->wall_ms;
    if (which == 5) return nd->scatter_ms;
    if (which == 6) return nd->used_rccl ? 1.0 : 0.0;
    if (rank < 0 || rank >= (int)nd->r.size()) return -1.0;
    const Rank &k = nd->r[rank];
    if (which == 1) return (double)k.streams;
    if (which == 2) return (double)k.in_bytes;
    if (which == 3) return k.kernel_ms;
    return -1.0;
}
