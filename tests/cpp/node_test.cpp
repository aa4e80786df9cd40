// node_test.cpp -- brx_node_* through nothing but include/brx.h: what a Rust -sys crate binds.
// usage: node_test <stream.compressed> <expected> <ranks> <copies>
// `ranks` virtual ranks on GPU 0 (or one rank per GPU when the machine has that many), `copies` of the stream as one batch in
// pinned host buffers (used in place by every rank's kernel), decoded with each deal; prints "OK ..." and exits 0 when every copy
// came out bit-exact with status 0.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/brx.h"

static std::vector<uint8_t> slurp(const char *path) {
    std::vector<uint8_t> v;
    FILE *f = fopen(path, "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize((size_t)n);
    if (n && fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
    fclose(f);
    return v;
}

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const std::vector<uint8_t> comp = slurp(argv[1]), want = slurp(argv[2]);
    const int ranks = atoi(argv[3]);
    const uint32_t n = (uint32_t)atoi(argv[4]);
    if (comp.empty() || ranks < 1 || n < 1) return 2;
    std::vector<int> devs((size_t)ranks, 0);
    brx_node *node = nullptr;
    if (brx_node_create(&node, devs.data(), ranks) != BRX_SUCCESS) {
        fprintf(stderr, "brx_node_create: %s\n", brx_last_error());
        return 1;
    }
    if (brx_node_size(node) != ranks || brx_node_ctx(node, ranks - 1) == nullptr) return 1;
    const size_t cap = (want.size() + 64 + 15) & ~(size_t)15;
    uint8_t *in = (uint8_t *)brx_host_alloc(comp.size() * n), *out = (uint8_t *)brx_host_alloc(cap * n);
    if (!in || !out) return 1;
    std::vector<uint64_t> in_off(n + 1), out_off(n + 1), out_len(n);
    std::vector<int32_t> status(n);
    for (uint32_t i = 0; i <= n; i++) {
        in_off[i] = (uint64_t)i * comp.size();
        out_off[i] = (uint64_t)i * cap;
    }
    for (uint32_t i = 0; i < n; i++) memcpy(in + in_off[i], comp.data(), comp.size());
    double ms[3] = {0, 0, 0};
    for (uint32_t deal = BRX_NODE_DEAL_RANGES; deal <= BRX_NODE_DEAL_SNAKE; deal++) {
        memset(out, 0xA5, cap * n);
        brx_node_opts o = {BRX_MEM_HOST | BRX_OPT_TIMING, deal, ranks, 0, nullptr};
        if (brx_node_decode_batch(node, in, in_off.data(), n, out, out_off.data(), out_len.data(), status.data(), &o) != BRX_SUCCESS) {
            fprintf(stderr, "brx_node_decode_batch: %s\n", brx_last_error());
            return 1;
        }
        if ((int)brx_node_last_timing(node, 0, 0) != ranks) return 1;
        uint32_t dealt = 0;
        for (int r = 0; r < ranks; r++) dealt += (uint32_t)brx_node_last_timing(node, 1, r);
        if (dealt != n) return 1;
        for (uint32_t i = 0; i < n; i++)
            if (status[i] != BRX_OK || out_len[i] != want.size() || memcmp(out + out_off[i], want.data(), want.size()) != 0) {
                fprintf(stderr, "deal %u stream %u: status %d (%s), %llu bytes\n", deal, i, status[i], brx_status_str(status[i]), (unsigned long long)out_len[i]);
                return 1;
            }
        ms[deal] = brx_node_last_timing(node, 4, 0);
    }
    printf("OK %u streams over %d ranks: %.2f / %.2f / %.2f ms (ranges / bytes / snake)\n", n, ranks, ms[0], ms[1], ms[2]);
    brx_host_free(in);
    brx_host_free(out);
    brx_node_destroy(node);
    return 0;
}
