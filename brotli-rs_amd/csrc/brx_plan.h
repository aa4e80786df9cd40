// brx_plan.h -- host-side arithmetic of launch plan B (brx_api.cpp launch()): which kernel instances run next to each other, with
// what grids, given how many streams the classification pre-pass found for each.  Pure C++ (no HIP): tests/cpp/plan_test.cpp checks
// its invariants on the CPU.
//
// A CU's 160 KiB of LDS is handed out in four parts of 40 KiB (one per SIMD; measured: profiles/r04_two_queues.txt,
// r04_residency.txt): 4 regular workgroups (10 KiB), 3 of level 1 (12.5), 2 of level 2 (20) or 1 of level 3 (40) per part -- and a
// dispatch that has run out of room does not move on when room appears elsewhere, its pending workgroups wait for the CU they are
// due on.  So every kernel gets a PERSISTENT grid of workgroups that all find room at once (the rest of its list comes through its
// ticket counter).  One class only: its own kernel, alone, at its full residency.  Several: level 1 joins level 2 -- parts fill
// without a gap only with sizes 10 / 20 / 40 KiB (12.5 + 2 x 10 leaves 7.5 unused, and a part that holds three regular workgroups
// holds nothing else) -- and the shares are in proportion to the parts each class would need for all its streams (about the same
// number of rounds for each); then the wide classes -- few streams, long ones -- get WHOLE rounds: 512 streams on 455 workgroups
// would leave 57 of them a second round behind everything else, so a share of at least 2/3 of a class becomes all of it, anything
// less the even split over its rounds.  The regular class takes what is left (at least a quarter of the chip while the wide ones
// can be halved): its workgroups go through many short streams, a few more or less resident change its time in proportion, not in
// steps.
#pragma once
#include <stdint.h>

#include <algorithm>

struct BrxPlanB {
    uint32_t grid[4];  // workgroups of the regular kernel / levels 1..3 (0 = not launched)
    uint32_t mask[4];  // lists a wider kernel decodes (bit j = list j = streams classified for level j + 1)
};

// n0 = streams of the regular kernel's queue that stay with it; cnt[1..3] = streams classified for levels 1..3; parts = 40-KiB LDS
// parts of the chip (4 per CU); max_grid = 16 per CU.
static inline BrxPlanB brx_plan_b(uint32_t n0, const uint32_t cnt[4], uint32_t parts, uint32_t max_grid) {
    const uint32_t per_cu = max_grid / 16u;
    BrxPlanB p = {{std::min(n0, max_grid), std::min(cnt[1], per_cu * 12u), std::min(cnt[2], per_cu * 8u), std::min(cnt[3], per_cu * 4u)},
                  {0u, 1u, 2u, 4u}};
    const int classes = (n0 != 0u) + (cnt[1] != 0u) + (cnt[2] != 0u) + (cnt[3] != 0u);
    if (classes <= 1) return p;
    const uint32_t m2 = cnt[1] + cnt[2];
    const double demand = n0 / 4.0 + m2 / 2.0 + cnt[3];
    const double sc = demand > (double)parts ? (double)parts / demand : 1.0; // every class's share of its streams
    auto whole_rounds = [&](uint32_t c) -> uint32_t {
        if (c == 0u) return 0u;
        const uint32_t rounds = (uint32_t)std::max(1.0, 1.0 / sc + 1.0 / 3.0); // 1 / share, rounded down from x.67
        return (c + rounds - 1u) / rounds;
    };
    uint32_t r3 = whole_rounds(cnt[3]), r2 = whole_rounds(m2);
    const uint32_t wide_parts_max = n0 ? parts - parts / 4u : parts;
    while (r3 + (r2 + 1u) / 2u > wide_parts_max && (r3 > 1u || r2 > 2u)) { // too much for the wide side: one more round each
        if (r3 > 1u) r3 = (r3 + 1u) / 2u;
        if (r2 > 2u) r2 = (r2 + 1u) / 2u;
    }
    const uint32_t left = parts - std::min(parts, r3 + (r2 + 1u) / 2u);
    p.grid[0] = std::min(std::min(n0, left * 4u), max_grid);
    if (n0 != 0u && p.grid[0] == 0u) p.grid[0] = 1u;
    p.grid[1] = 0u;
    p.grid[2] = r2;
    p.grid[3] = r3;
    p.mask[2] = 3u; // lists 0 and 1
    return p;
}
