// Stress of the Read facade's locking (round 6: queue lock + leading reader + per-stream condition variables): T threads, each K
// times one of -- a stream read to its end in chunks of random size; a stream PULLED through brx_stream_new_reader (the bounded path:
// it takes the context's lock slice by slice, between the facade's batches); two streams made, one freed unread, the other read; a
// stream made and freed at once; a stream read halfway and freed.  The files are given as pairs <compressed> <expected or "-status">.
// Every byte delivered is compared; exits 1 on any mismatch.  usage: stream_mix <threads> <iterations> <compressed> <expected> ...
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/brx.h"

struct Item {
    std::vector<uint8_t> comp, want;
    int64_t status = 0; // > 0: the stream must end with -status after a prefix of `want` (empty want: prefix not checked)
};
static std::vector<uint8_t> slurp(const char *p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
struct Src {
    const std::vector<uint8_t> *data;
    size_t at;
    std::mt19937 *rng;
};
static size_t pull(void *user, uint8_t *buf, size_t cap) {
    Src *s = (Src *)user;
    size_t k = std::min(cap, s->data->size() - s->at);
    if (k > 1 && ((*s->rng)() & 3u) == 0u) k = 1 + (*s->rng)() % k; // short reads
    memcpy(buf, s->data->data() + s->at, k);
    s->at += k;
    return k;
}

int main(int argc, char **argv) {
    if (argc < 5 || (argc - 3) % 2) return 2;
    const int T = atoi(argv[1]), K = atoi(argv[2]);
    std::vector<Item> items;
    for (int i = 3; i + 1 < argc; i += 2) {
        Item it;
        it.comp = slurp(argv[i]);
        if (argv[i + 1][0] == '-') it.status = atoi(argv[i + 1] + 1);
        else it.want = slurp(argv[i + 1]);
        items.push_back(std::move(it));
    }
    brx_ctx *ctx = nullptr;
    if (brx_ctx_create(&ctx, 0) != BRX_SUCCESS) return 1;
    std::atomic<int> bad{0}, done{0};
    auto read_all = [&](brx_stream *s, const Item &it, std::mt19937 &rng, size_t stop_after) -> bool {
        std::vector<uint8_t> buf(1 << 20);
        size_t at = 0;
        for (;;) {
            const size_t ask = 1 + rng() % buf.size();
            const int64_t n = brx_stream_read(s, buf.data(), ask);
            if (n < 0) return it.status > 0 && n == -it.status;
            if (n == 0) return it.status == 0 && at == it.want.size();
            if (it.status == 0 || !it.want.empty()) {
                if (at + (size_t)n > it.want.size() && it.status == 0) return false;
                const size_t m = std::min((size_t)n, it.want.size() > at ? it.want.size() - at : 0);
                if (m && memcmp(buf.data(), it.want.data() + at, m) != 0) return false;
            }
            at += (size_t)n;
            if (at >= stop_after) return true;
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            std::mt19937 rng(1234u + (unsigned)t);
            for (int k = 0; k < K; k++) {
                const Item &it = items[rng() % items.size()];
                const unsigned what = rng() % 8u;
                bool ok = true;
                if (what <= 2) {
                    brx_stream *s = brx_stream_new(ctx, it.comp.data(), it.comp.size());
                    ok = s && read_all(s, it, rng, (size_t)-1);
                    brx_stream_free(s);
                } else if (what == 3) {
                    Src src{&it.comp, 0, &rng};
                    brx_stream *s = brx_stream_new_reader(ctx, pull, &src);
                    ok = s && read_all(s, it, rng, (size_t)-1);
                    brx_stream_free(s);
                } else if (what == 4) {
                    const Item &other = items[rng() % items.size()];
                    brx_stream *a = brx_stream_new(ctx, other.comp.data(), other.comp.size());
                    brx_stream *b = brx_stream_new(ctx, it.comp.data(), it.comp.size());
                    if (rng() & 1u) { brx_stream_free(a); a = nullptr; }
                    ok = b && read_all(b, it, rng, (size_t)-1);
                    brx_stream_free(b);
                    brx_stream_free(a); // (unread, possibly decoded by somebody's batch in the meantime)
                } else if (what == 5) {
                    brx_stream_free(brx_stream_new(ctx, it.comp.data(), it.comp.size()));
                } else if (what == 6) {
                    brx_stream *s = brx_stream_new(ctx, it.comp.data(), it.comp.size());
                    ok = s && read_all(s, it, rng, 1 + rng() % 5000);
                    brx_stream_free(s);
                } else {
                    brx_stream *s = brx_stream_new_bounded(ctx, it.comp.data(), it.comp.size());
                    ok = s && read_all(s, it, rng, (size_t)-1);
                    brx_stream_free(s);
                }
                if (!ok) bad++;
                done++;
            }
        });
    for (auto &x : th) x.join();
    printf("stream_mix: %d threads x %d: %d done, %d wrong; facade batches %.0f with %.0f streams\n", T, K, done.load(), bad.load(), brx_last_timing(ctx, 14),
           brx_last_timing(ctx, 15));
    brx_ctx_destroy(ctx);
    return bad.load() ? 1 : 0;
}
