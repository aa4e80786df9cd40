// diag_main.cpp -- standalone driver of the C ABI (no Python, no torch): decode one file, print status.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/brx.h"

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s <file.compressed> <capacity> [copies]\n", argv[0]);
        return 2;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    std::vector<uint8_t> in;
    uint8_t buf[65536];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) in.insert(in.end(), buf, buf + k);
    fclose(f);
    size_t cap = strtoull(argv[2], nullptr, 10);
    unsigned copies = argc > 3 ? atoi(argv[3]) : 1;
    fprintf(stderr, "[diag] input %zu B, cap %zu, copies %u\n", in.size(), cap, copies);
    brx_ctx *ctx = nullptr;
    int rc = brx_ctx_create(&ctx, 0);
    fprintf(stderr, "[diag] ctx_create rc=%d %s\n", rc, rc ? brx_last_error() : "");
    if (rc) return 1;
    std::vector<uint8_t> all;
    std::vector<uint64_t> in_off(copies + 1), out_off(copies + 1), out_len(copies);
    std::vector<int32_t> st(copies, -1);
    for (unsigned i = 0; i < copies; i++) {
        in_off[i] = all.size();
        all.insert(all.end(), in.begin(), in.end());
        out_off[i] = (uint64_t)i * cap;
    }
    in_off[copies] = all.size();
    out_off[copies] = (uint64_t)copies * cap;
    std::vector<uint8_t> out((size_t)copies * cap + 16);
    brx_opts o = {BRX_MEM_HOST | BRX_OPT_TIMING, 0, nullptr};
    const unsigned reps = argc > 4 ? atoi(argv[4]) : 1; // host-pointer path: the wall time includes H2D and D2H
    for (unsigned r = 0; r < reps; r++) {
        auto t0 = std::chrono::steady_clock::now();
        rc = brx_decode_batch(ctx, all.data(), in_off.data(), copies, out.data(), out_off.data(), out_len.data(), st.data(), &o);
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[diag] decode rc=%d %s wall %.3f ms kernel %.3f ms\n", rc, rc ? brx_last_error() : "",
                std::chrono::duration<double, std::milli>(t1 - t0).count(), brx_last_timing(ctx, 1));
    }
    uint32_t h = 2166136261u;
    for (size_t i = 0; i < out_len[0] && st[0] == 0; i++) h = (h ^ out[i]) * 16777619u;
    printf("status=%d out_len=%llu fnv=%08x (%s)\n", st[0], (unsigned long long)out_len[0], h, brx_status_str(st[0]));
    brx_ctx_destroy(ctx);
    return 0;
}
