// CPU test of launch plan B's grid arithmetic (brotli-rs_amd/csrc/brx_plan.h): invariants over a sweep of class counts.
#include <cstdio>
#include <cstdlib>

#include "../../brotli-rs_amd/csrc/brx_plan.h"

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails++ < 10) { printf("FAIL %s: ", #c); printf(__VA_ARGS__); printf("\n"); } } } while (0)

int main() {
    const uint32_t max_grid = 4096, parts = 1024;
    const uint32_t vals[] = {0, 1, 2, 3, 7, 57, 455, 512, 513, 1000, 1024, 1025, 2048, 3072, 4096, 5000, 9999, 65536, 262144};
    uint64_t cases = 0;
    for (uint32_t n0 : vals) for (uint32_t c1 : vals) for (uint32_t c2 : vals) for (uint32_t c3 : vals) {
        if (n0 + c1 + c2 + c3 == 0) continue;
        const uint32_t cnt[4] = {0, c1, c2, c3};
        const BrxPlanB p = brx_plan_b(n0, cnt, parts, max_grid);
        cases++;
        const int classes = (n0 != 0) + (c1 != 0) + (c2 != 0) + (c3 != 0);
        // every stream is reachable: a class with streams has a kernel whose mask covers its list
        uint32_t covered = 0;
        for (int k = 1; k < 4; k++) if (p.grid[k]) covered |= p.mask[k];
        CHECK(!c1 || (covered & 1u), "class 1 uncovered n0=%u c=%u,%u,%u", n0, c1, c2, c3);
        CHECK(!c2 || (covered & 2u), "class 2 uncovered n0=%u c=%u,%u,%u", n0, c1, c2, c3);
        CHECK(!c3 || (covered & 4u), "class 3 uncovered n0=%u c=%u,%u,%u", n0, c1, c2, c3);
        CHECK(!n0 || p.grid[0] >= 1, "regular class without a workgroup n0=%u c=%u,%u,%u", n0, c1, c2, c3);
        CHECK(n0 || p.grid[0] == 0, "regular kernel without streams");
        // no list is decoded by two kernels
        CHECK((p.grid[1] ? p.mask[1] : 0u) + (p.grid[2] ? p.mask[2] : 0u) + (p.grid[3] ? p.mask[3] : 0u) == covered, "a list in two kernels");
        // never more workgroups than streams of the lists a kernel decodes, nor than its instance's residency
        const uint32_t s1 = (p.mask[1] & 1u ? c1 : 0), s2 = (p.mask[2] & 1u ? c1 : 0) + (p.mask[2] & 2u ? c2 : 0);
        CHECK(p.grid[0] <= n0 && p.grid[0] <= max_grid, "g0 %u", p.grid[0]);
        CHECK(p.grid[1] <= s1 && p.grid[1] <= 3072, "g1 %u", p.grid[1]);
        CHECK(p.grid[2] <= s2 && p.grid[2] <= 2048, "g2 %u of %u", p.grid[2], s2);
        CHECK(p.grid[3] <= c3 && p.grid[3] <= 1024, "g3 %u", p.grid[3]);
        if (classes > 1) {
            // what is launched together fits the chip's LDS parts (the regular kernel's one forced workgroup aside)
            const uint32_t used = (p.grid[0] + 3) / 4 + (p.grid[1] + 2) / 3 + (p.grid[2] + 1) / 2 + p.grid[3];
            CHECK(used <= parts + 1, "parts %u > %u: n0=%u c=%u,%u,%u -> g=%u,%u,%u,%u", used, parts, n0, c1, c2, c3, p.grid[0], p.grid[1], p.grid[2], p.grid[3]);
            CHECK(p.grid[1] == 0, "level 1 on its own in a mixed batch");
            // the regular class keeps at least a quarter of the chip when it could use it
            if (n0 >= max_grid / 4) CHECK(p.grid[0] >= max_grid / 4 - 4, "regular class squeezed: g0=%u n0=%u c=%u,%u,%u", p.grid[0], n0, c1, c2, c3);
        } else {
            // one class: its own kernel at its full residency
            CHECK(p.grid[0] == (n0 < max_grid ? n0 : max_grid), "single regular");
            CHECK(p.grid[1] == (c1 < 3072 ? c1 : 3072) && p.grid[2] == (c2 < 2048 ? c2 : 2048) && p.grid[3] == (c3 < 1024 ? c3 : 1024), "single wide");
        }
    }
    // the two measured batches
    { const uint32_t cnt[4] = {0, 0, 512, 0}; BrxPlanB p = brx_plan_b(3584, cnt, parts, max_grid); CHECK(p.grid[2] == 512 && p.grid[0] == 3072, "mixed_all: %u %u", p.grid[0], p.grid[2]); }
    { const uint32_t cnt[4] = {0, 1024, 0, 0}; BrxPlanB p = brx_plan_b(3072, cnt, parts, max_grid); CHECK(p.grid[1] == 0 && p.grid[2] > 0 && p.mask[2] == 3u, "level 1 joins level 2"); }
    printf("%llu cases, %d failures\n", (unsigned long long)cases, fails);
    return fails ? 1 : 0;
}
