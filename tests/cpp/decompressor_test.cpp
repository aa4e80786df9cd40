// Mirror of the reference's integration tests (tests/lib.rs) through the C++ Decompressor facade.
// usage: decompressor_test <stream> <expected|-> [expected error substring]
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>

#include "../../brotli-rs_amd/host/decompressor.hpp"

static std::vector<uint8_t> slurp(const char *p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), {});
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    std::vector<uint8_t> in = slurp(argv[1]);
    brotli::Decompressor<brotli::SliceReader> d(brotli::SliceReader(in.data(), in.size()));
    try {
        std::vector<uint8_t> out = d.read_to_end();
        if (argc > 3) { printf("FAIL: expected error '%s', decoded %zu bytes\n", argv[3], out.size()); return 1; }
        std::vector<uint8_t> exp = slurp(argv[2]);
        if (out != exp) { printf("FAIL: output differs (%zu vs %zu bytes)\n", out.size(), exp.size()); return 1; }
        uint8_t b;
        if (d.read(&b, 1) != 0) { printf("FAIL: read after end returned data\n"); return 1; }
        printf("OK %zu bytes\n", out.size());
        return 0;
    } catch (const brotli::InvalidData &e) {
        if (argc > 3 && strstr(e.what(), argv[3])) { printf("OK error: %s\n", e.what()); return 0; }
        printf("FAIL: unexpected InvalidData: %s\n", e.what());
        return 1;
    }
}
