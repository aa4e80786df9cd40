// brx_walk -- the reference's file walker (src/main.rs:49-70: every regular file of a directory whose name ends with
// "compressed" is decompressed, its path, output length and result printed) on the batched decoder: all files of the
// directory are read straight into ONE pinned host buffer (brx_host_alloc), decoded as ONE batch through the C ABI's
// host-pointer path (pinned staging, chunked copy in / decode / copy out on HIP streams, brx.h) into a pinned output
// arena, and reported in the reference's shape.  Output sizes are not known in advance (a Brotli stream does not carry
// its size): capacities start at 8 x the compressed size and every stream that reports status 25 (capacity too small) is
// decoded again, alone with the others that did, at 8 x the capacity -- the size-discovery retry of SURVEY 8f-2.
//
//   brx_walk <dir> [--suffix S] [--check] [--out DIR] [--quiet] [--gpus N | --ranks d0,d1,...]
//     --suffix S   file-name ending to look for (default "compressed", as the reference)
//     --check      compare each output with the file named by the part before ".compressed" when it exists
//                  (the layout of the reference's data/ directory); exit status 1 on any mismatch
//     --out DIR    write each output to DIR/<name>.out
//     --quiet      only the summary
//     --gpus N     (round 6) the batch over the first N GPUs of the machine through brx_node_decode_batch -- one process, one host
//                  thread per GPU, files dealt by compressed size (snake deal: files differ), every GPU reads / writes the pinned
//                  buffers in place, no exchange between GPUs; --ranks 0,0,1 names the device of every rank instead (a device may
//                  repeat: several ranks on one GPU)
// Host code above the C ABI only: no HIP calls here, no decoding on the CPU (there is none in the library).
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/brx.h"

namespace {

struct Entry {
    std::string path, name;
    uint64_t size = 0;
};

bool ends_with(const std::string &s, const std::string &suffix) {
    return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
}

bool read_file(const std::string &path, uint8_t *dst, uint64_t size) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    const size_t got = fread(dst, 1, size, f);
    fclose(f);
    return got == size;
}

uint64_t align16(uint64_t v) { return (v + 15u) & ~(uint64_t)15u; }

}  // namespace

int main(int argc, char **argv) {
    std::string dir, suffix = "compressed", out_dir;
    bool check = false, quiet = false;
    std::vector<int> ranks;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--suffix" && i + 1 < argc) suffix = argv[++i];
        else if (a == "--gpus" && i + 1 < argc) { const int g = atoi(argv[++i]); for (int r = 0; r < g; r++) ranks.push_back(r); }
        else if (a == "--ranks" && i + 1 < argc) { for (const char *p = argv[++i]; *p;) { ranks.push_back(atoi(p)); while (*p && *p != ',') p++; if (*p == ',') p++; } }
        else if (a == "--out" && i + 1 < argc) out_dir = argv[++i];
        else if (a == "--check") check = true;
        else if (a == "--quiet") quiet = true;
        else if (dir.empty()) dir = a;
        else { fprintf(stderr, "usage: %s <dir> [--suffix S] [--check] [--out DIR] [--quiet]\n", argv[0]); return 2; }
    }
    if (dir.empty()) { fprintf(stderr, "usage: %s <dir> [--suffix S] [--check] [--out DIR] [--quiet]\n", argv[0]); return 2; }

    // the walk: regular files only, no recursion (src/main.rs:52-58), sorted for a stable report
    std::vector<Entry> files;
    DIR *d = opendir(dir.c_str());
    if (!d) { perror(dir.c_str()); return 2; }
    while (dirent *e = readdir(d)) {
        Entry f;
        f.name = e->d_name;
        f.path = dir + "/" + f.name;
        struct stat st;
        if (stat(f.path.c_str(), &st) != 0 || !S_ISREG(st.st_mode) || !ends_with(f.name, suffix)) continue;
        f.size = (uint64_t)st.st_size;
        files.push_back(f);
    }
    closedir(d);
    std::sort(files.begin(), files.end(), [](const Entry &a, const Entry &b) { return a.name < b.name; });
    const uint32_t n = (uint32_t)files.size();
    if (n == 0) { printf("no files ending with \"%s\" in %s\n", suffix.c_str(), dir.c_str()); return 0; }

    brx_ctx *ctx = nullptr;
    brx_node *node = nullptr;
    if (!ranks.empty()) {
        if (brx_node_create(&node, ranks.data(), (int)ranks.size()) != BRX_SUCCESS) { fprintf(stderr, "brx_node_create: %s\n", brx_last_error()); return 3; }
    } else if (brx_ctx_create(&ctx, 0) != BRX_SUCCESS) { fprintf(stderr, "brx_ctx_create: %s\n", brx_last_error()); return 3; }

    // ingest: every file straight into one pinned buffer
    std::vector<uint64_t> in_off(n + 1, 0);
    for (uint32_t i = 0; i < n; i++) in_off[i + 1] = in_off[i] + files[i].size;
    uint8_t *in = (uint8_t *)brx_host_alloc(std::max<uint64_t>(in_off[n], 16));
    if (!in) { fprintf(stderr, "brx_host_alloc(%llu) failed\n", (unsigned long long)in_off[n]); return 3; }
    for (uint32_t i = 0; i < n; i++)
        if (!read_file(files[i].path, in + in_off[i], files[i].size)) { fprintf(stderr, "cannot read %s\n", files[i].path.c_str()); return 3; }

    // decode; streams whose slot was too small go round again with 8 x the capacity
    std::vector<int32_t> status(n, -1);
    std::vector<uint64_t> out_len(n, 0), cap(n), slot_off(n, 0);
    std::vector<uint8_t *> arenas;
    std::vector<uint32_t> slot_arena(n, 0);
    for (uint32_t i = 0; i < n; i++) cap[i] = align16(std::max<uint64_t>(files[i].size * 8u, 65536u));
    std::vector<uint32_t> todo(n);
    for (uint32_t i = 0; i < n; i++) todo[i] = i;
    double decode_ms = 0;
    int rounds = 0;
    while (!todo.empty()) {
        rounds++;
        const uint32_t k = (uint32_t)todo.size();
        std::vector<uint64_t> r_in_off(k + 1, 0), r_out_off(k + 1, 0), r_len(k, 0);
        std::vector<int32_t> r_st(k, -1);
        for (uint32_t j = 0; j < k; j++) r_out_off[j + 1] = r_out_off[j] + cap[todo[j]];
        uint8_t *arena = (uint8_t *)brx_host_alloc(std::max<uint64_t>(r_out_off[k], 16));
        if (!arena) { fprintf(stderr, "brx_host_alloc(%llu) failed\n", (unsigned long long)r_out_off[k]); return 3; }
        arenas.push_back(arena);
        // the streams of a retry round are not contiguous in `in`: gather them (first round: the buffer as it is)
        const uint8_t *r_in = in;
        uint8_t *gathered = nullptr;
        if (k == n) {
            r_in_off = in_off;
        } else {
            for (uint32_t j = 0; j < k; j++) r_in_off[j + 1] = r_in_off[j] + files[todo[j]].size;
            gathered = (uint8_t *)brx_host_alloc(std::max<uint64_t>(r_in_off[k], 16));
            if (!gathered) { fprintf(stderr, "brx_host_alloc failed\n"); return 3; }
            for (uint32_t j = 0; j < k; j++) memcpy(gathered + r_in_off[j], in + in_off[todo[j]], files[todo[j]].size);
            r_in = gathered;
        }
        brx_opts o = {BRX_MEM_HOST, 0, nullptr};
        brx_node_opts no = {BRX_MEM_HOST, BRX_NODE_DEAL_SNAKE, (int32_t)ranks.size(), 0, nullptr};
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = node ? brx_node_decode_batch(node, r_in, r_in_off.data(), k, arena, r_out_off.data(), r_len.data(), r_st.data(), &no)
                            : brx_decode_batch(ctx, r_in, r_in_off.data(), k, arena, r_out_off.data(), r_len.data(), r_st.data(), &o);
        decode_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (gathered) brx_host_free(gathered);
        if (rc != BRX_SUCCESS) { fprintf(stderr, "brx_decode_batch: %s\n", brx_last_error()); return 3; }
        std::vector<uint32_t> again;
        for (uint32_t j = 0; j < k; j++) {
            const uint32_t i = todo[j];
            status[i] = r_st[j];
            out_len[i] = r_len[j];
            slot_arena[i] = (uint32_t)arenas.size() - 1;
            slot_off[i] = r_out_off[j];
            if (r_st[j] == 25 && cap[i] < ((uint64_t)1 << 32) - 4096) {  // capacity too small: grow (the per-stream limit is 4 GiB - 256 B)
                cap[i] = std::min<uint64_t>(align16(cap[i] * 8u), ((uint64_t)1 << 32) - 4096);
                again.push_back(i);
            }
        }
        todo.swap(again);
    }

    // the report, in the reference's shape (src/main.rs:60-65)
    uint64_t total_in = in_off[n], total_out = 0;
    uint32_t n_ok = 0, n_err = 0, n_mismatch = 0, n_checked = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint8_t *out = arenas[slot_arena[i]] + slot_off[i];
        const bool ok = status[i] == 0;
        ok ? n_ok++ : n_err++;
        total_out += out_len[i];
        std::string verdict;
        if (check) {
            const size_t cut = files[i].name.find(".compressed");
            const std::string expect_path = dir + "/" + files[i].name.substr(0, cut);
            struct stat st;
            if (cut != std::string::npos && cut > 0 && stat(expect_path.c_str(), &st) == 0 && S_ISREG(st.st_mode)) {
                if (ok) {
                    std::vector<uint8_t> want((size_t)st.st_size);
                    const bool same = read_file(expect_path, want.data(), want.size()) && want.size() == out_len[i] &&
                                      (want.empty() || memcmp(want.data(), out, want.size()) == 0);
                    n_checked++;
                    if (!same) n_mismatch++;
                    verdict = same ? "check = identical to " + expect_path : "check = DIFFERS from " + expect_path;
                } else {
                    verdict = "check = (an error; expected file exists: " + expect_path + ")";
                }
            } else {
                verdict = "check = (no expected file)";
            }
        }
        if (!quiet) {
            printf("\"%s\":\n", files[i].path.c_str());
            printf("output length = %llu\n", (unsigned long long)out_len[i]);
            if (ok) printf("res = Ok(%llu)\n", (unsigned long long)out_len[i]);
            else printf("res = Err(\"%s\")  [status %d]\n", brx_status_str(status[i]), status[i]);
            if (check) printf("%s\n", verdict.c_str());
            printf("===========\n\n");
        }
        if (!out_dir.empty() && ok) {
            const std::string p = out_dir + "/" + files[i].name + ".out";
            FILE *f = fopen(p.c_str(), "wb");
            if (!f || fwrite(out, 1, out_len[i], f) != out_len[i]) { fprintf(stderr, "cannot write %s\n", p.c_str()); return 3; }
            fclose(f);
        }
    }
    printf("%u files, %u decoded, %u errors; %llu B in, %llu B out; %d batch call(s), %.2f ms in %s "
           "(host buffers: copies included, %.1f MB/s decompressed)",
           n, n_ok, n_err, (unsigned long long)total_in, (unsigned long long)total_out, rounds, decode_ms,
           node ? "brx_node_decode_batch" : "brx_decode_batch", decode_ms > 0 ? total_out / decode_ms / 1e3 : 0.0);
    if (node) printf("; %d ranks", (int)ranks.size());
    if (check) printf("; %u compared with their expected files, %u differ", n_checked, n_mismatch);
    printf("\n");
    for (uint8_t *a : arenas) brx_host_free(a);
    brx_host_free(in);
    if (node) brx_node_destroy(node);
    brx_ctx_destroy(ctx);
    return n_mismatch ? 1 : 0;
}
