// The reference's usage from a multi-threaded host: T threads, each K times { Decompressor::new(bytes) -> read to the end -> drop }
// (tests/lib.rs's pattern on every thread of a server), against nothing but include/brx.h.  Streams made while a batch is running
// queue up and the next read decodes all of them at once: the threads coalesce into batches by themselves.  Prints the rate, the
// number of batches the facade launched and their mean size; exits 1 on any wrong byte.
// usage: stream_threads <compressed> <expected> <threads> <iterations>        every stream must decode to <expected>
//        stream_threads <compressed> - <threads> <iterations> <status>    every stream must end with -<status> (a cut / corrupt file)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <thread>
#include <vector>

#include "../../include/brx.h"

static std::vector<uint8_t> slurp(const char *p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const bool fails = strcmp(argv[2], "-") == 0;
    const std::vector<uint8_t> comp = slurp(argv[1]), want = fails ? std::vector<uint8_t>() : slurp(argv[2]);
    const int T = atoi(argv[3]), K = atoi(argv[4]);
    const int64_t want_status = fails && argc > 5 ? atoi(argv[5]) : 0;
    brx_ctx *ctx = nullptr;
    if (brx_ctx_create(&ctx, 0) != BRX_SUCCESS) {
        fprintf(stderr, "brx_ctx_create: %s\n", brx_last_error());
        return 1;
    }
    { // warm-up: one stream alone (allocations, first launch)
        brx_stream *s = brx_stream_new(ctx, comp.data(), comp.size());
        std::vector<uint8_t> buf(1 << 16);
        while (brx_stream_read(s, buf.data(), buf.size()) > 0) {}
        brx_stream_free(s);
    }
    const double b0 = brx_last_timing(ctx, 14), n0 = brx_last_timing(ctx, 15);
    std::atomic<int> bad{0};
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&] {
            std::vector<uint8_t> got(fails ? comp.size() * 64 + 65536 : want.size() + 4096);
            for (int k = 0; k < K; k++) {
                brx_stream *s = brx_stream_new(ctx, comp.data(), comp.size());
                if (!s) { bad++; return; }
                size_t at = 0;
                for (;;) {
                    const int64_t n = brx_stream_read(s, got.data() + at, got.size() - at);
                    if (n < 0) {
                        if (!fails || n != -want_status) bad++;
                        break;
                    }
                    if (n == 0) {
                        if (fails) bad++;
                        break;
                    }
                    at += (size_t)n;
                }
                if (!fails && (at != want.size() || memcmp(got.data(), want.data(), at) != 0)) bad++;
                brx_stream_free(s);
            }
        });
    for (auto &x : th) x.join();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const double batches = brx_last_timing(ctx, 14) - b0, streams = brx_last_timing(ctx, 15) - n0;
    printf("%d threads x %d: %d streams of %zu B in %.1f ms = %.0f MB/s; %.0f batches, %.1f streams each; %d wrong\n", T, K, T * K, want.size(), ms,
           (double)T * K * want.size() / ms / 1e3, batches, batches > 0 ? streams / batches : 0.0, bad.load());
    brx_ctx_destroy(ctx);
    return bad.load() ? 1 : 0;
}
