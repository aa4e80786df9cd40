// The pulled reader (brx_stream_new_reader through brotli::Decompressor<R>, host/decompressor.hpp) on a stream of any length made
// in constant memory: prefix + unit x K + final (tests/craft.py periodic_stream_parts) must decode to unit_output x K -- the
// reference's Decompressor<R: Read> over a BufReader (src/lib.rs:398-410, src/bitreader/mod.rs:21-53).
// usage: stream_reader_test <prefix> <unit> <final> <unit_output> <K> [cut_bytes]
//        cut_bytes > 0: the input ends after that many bytes (expects UnexpectedEOF after the prefix of the output)
// prints: OK bytes <n> device_peak_MiB <x> host_hwm_MiB <y> (before the stream: <z>) seconds <t> MB_per_s <r>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>

#include "../../brotli-rs_amd/host/decompressor.hpp"

static std::vector<uint8_t> slurp(const char *p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), {});
}

struct PeriodicReader { // a "Read" over prefix + unit x K + final, nothing materialised
    std::vector<uint8_t> prefix, unit, fin;
    uint64_t K, at = 0, limit;
    uint64_t total() const { return prefix.size() + unit.size() * K + fin.size(); }
    size_t read(uint8_t *buf, size_t len) {
        const uint64_t end = limit < total() ? limit : total();
        size_t k = 0;
        if (len > 300007) len = 300007; // ragged chunks
        while (k < len && at < end) {
            const uint8_t *src;
            uint64_t room;
            if (at < prefix.size()) { src = prefix.data() + at; room = prefix.size() - at; }
            else if (at < prefix.size() + unit.size() * K) { uint64_t o = (at - prefix.size()) % unit.size(); src = unit.data() + o; room = unit.size() - o; }
            else { uint64_t o = at - prefix.size() - unit.size() * K; src = fin.data() + o; room = fin.size() - o; }
            uint64_t n = room < len - k ? room : len - k;
            if (n > end - at) n = end - at;
            memcpy(buf + k, src, n);
            k += n; at += n;
        }
        return k;
    }
};

static size_t device_free() { size_t f = 0, t = 0; (void)hipMemGetInfo(&f, &t); return f; }
static double host_hwm_mib() {
    std::ifstream f("/proc/self/status");
    std::string line;
    while (std::getline(f, line)) if (line.rfind("VmHWM:", 0) == 0) return atof(line.c_str() + 6) / 1024.0;
    return -1;
}

int main(int argc, char **argv) {
    if (argc < 6) return 2;
    PeriodicReader r{slurp(argv[1]), slurp(argv[2]), slurp(argv[3]), (uint64_t)atoll(argv[5]), 0, ~0ull};
    const std::vector<uint8_t> uo = slurp(argv[4]);
    const uint64_t cut = argc > 6 ? (uint64_t)atoll(argv[6]) : 0;
    if (cut) r.limit = cut;
    const uint64_t K = r.K;
    (void)brotli::default_context();
    { // (one small decode first: the baseline is a context that has launched before -- code objects loaded, runtime pools made)
        static const uint8_t tiny[] = {0x0b, 0x00, 0x80, 0x58, 0x03}; // "X": tests/lib.rs should_decompress_to_string style vector (1 byte)
        brotli::Decompressor<brotli::SliceReader> w(brotli::SliceReader(tiny, sizeof tiny));
        try { (void)w.read_to_end(); } catch (...) {}
        static std::vector<uint8_t> big(5u << 20, 0);  // ... and one through the bounded path (forced by its size; fails at once)
        brotli::Decompressor<brotli::SliceReader> w2(brotli::SliceReader(big.data(), big.size()));
        try { (void)w2.read_to_end(); } catch (...) {}
    }
    const size_t free0 = device_free();
    const double hwm0 = host_hwm_mib();
    size_t free_min = free0;
    brotli::Decompressor<PeriodicReader> d(std::move(r));
    std::vector<uint8_t> buf(1 << 20);
    uint64_t got = 0;
    const auto t0 = std::chrono::steady_clock::now();
    try {
        for (;;) {
            size_t k = d.read(buf.data(), buf.size());
            if (k == 0) break;
            for (size_t i = 0; i < k;) { // compare with unit_output repeated
                const size_t o = (size_t)((got + i) % uo.size()), n = uo.size() - o < k - i ? uo.size() - o : k - i;
                if (memcmp(buf.data() + i, uo.data() + o, n) != 0) { printf("FAIL: output differs near byte %llu\n", (unsigned long long)(got + i)); return 1; }
                i += n;
            }
            got += k;
            const size_t f = device_free();
            if (f < free_min) free_min = f;
        }
    } catch (const brotli::InvalidData &e) {
        if (cut && e.status == 24) { printf("OK error after %llu bytes: %s\n", (unsigned long long)got, e.what()); return got > 0 ? 0 : 1; }
        printf("FAIL: InvalidData after %llu bytes: %s\n", (unsigned long long)got, e.what());
        return 1;
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (cut || got != uo.size() * K) { printf("FAIL: %llu bytes, expected %llu\n", (unsigned long long)got, (unsigned long long)(uo.size() * K)); return 1; }
    printf("OK bytes %llu device_peak_MiB %.1f host_hwm_MiB %.1f (before the stream: %.1f ) seconds %.2f MB_per_s %.1f\n", (unsigned long long)got,
           (free0 - free_min) / 1048576.0, host_hwm_mib(), hwm0, sec, got / sec / 1e6);
    return 0;
}
