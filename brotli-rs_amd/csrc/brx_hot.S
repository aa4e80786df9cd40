// brx_hot.S -- the command loop of one compressed meta-block, hand-written for gfx950 (CDNA4).  Round-2 rewrite.
//
// Reference states DataMetaBlockBegin .. CopyLiterals (src/lib.rs:2003-2141): insert&copy symbol, extra bits,
// context-modelled literals, distance symbol / last-distance ring, window copy, dictionary word.  One wavefront = one
// stream; all decoder state is wave-uniform (SGPRs for what steers control flow, "uniform VGPRs" -- the same value in
// every lane -- for the bit window and the arithmetic, see TAKE below); the 64 lanes are used as (a) a 256-byte staging
// buffer of the compressed input (v_readlane feeds the 64-bit bit window), (b) comparators of the canonical prefix-code
// lookup (lane L holds the left-aligned exclusive upper bound of the length-L codes: one v_cmp + s_ff1 = code length),
// (c) byte movers AND a queue of copies in flight: a copy of <= 64 bytes takes the next free lanes of ONE pending
// register (its load is issued under an EXEC mask of just those lanes), so several back-references -- far ones read
// the stream's own output in HBM -- are in flight at once; they land in the LDS ring with a single masked ds_write when
// their bytes are needed (literal context, an overlapping source, a flush, 64 lanes used up).
//
// What one single wave pays per instruction is ~5 cycles of issue plus ~50 per dependent LDS / scalar-cache round trip
// (tools/ubench/lat.hip), so the loop is written for few instructions and few dependent trips per symbol:
//   * the symbol entries of the prefix-code tables carry what the loop needs next (prepare_fast_tables() in
//     brx_kernels.hip): literal = byte | context info, insert&copy = offset of the symbol's record (one s_load_dwordx2:
//     both bases, both extra-bit counts, implicit-distance flag), distance = extra-bit count | base (or a ring code);
//   * limit and base of a code length share one word (one ds_read_b32 per lane loads a tree; SPLIT_TREE makes the pair the
//     lookups work on);
//   * the literal context id is two masks of the previous entries' info bytes; a one-tree meta-block keeps its literal
//     tree in registers and skips contexts altogether;
//   * the common path of a command falls through; everything unusual is out of line.
//
// This file is preprocessed (register names) and pasted into ONE asm statement of brx_kernels.hip
// (asm_commands()).  It talks to the C++ segments only through the parked state in LDS (Lds::st, Lds::mbw):
//   entry : always at resume point R1 (insert_len / copy_len / implicit_zero of the current command known)
//   exit  : mbw[MBW_EXIT] = 0 (R0: insert&copy symbol due), 1 (R1), 2 (R2: distance of the current command known).
// Everything unusual leaves through one of those points and is handled by generic_commands() in C++, which runs one
// command and hands back: copies longer than 512 bytes (8 191 when the source is in the ring), any error (the C++ side re-decodes and
// raises it), the dword the dispatcher names in mbw[MBW_WSAFE] (resumable decode: END_MARGIN dwords in front of the end of the
// input, so that every bit consumed below is a real bit; batches: see "speculative end" in brx_kernels.hip), a ragged first flush
// block, extra-bit fields wider than the window, and the block switches the loop does not take itself: a block type / block count
// code that is not a complete general code, literal block types that differ in context mode.  (Block switches of all three
// categories ARE taken inside the loop -- .Lswitch, .Lsw_L, .Lx_dist_switch, .Lx_r0_switch; the sentence that said
// otherwise here until round 6 was stale, and VERDICT r5 read it.)
//
// Preconditions (the HC_START call of generic_commands() sets mbw[MBW_ASM] and rewrites the tables): every literal and
// distance tree is a complete general code or a one-symbol code, every insert&copy tree a complete general code, all
// of them and the context maps resident in LDS table memory, <= 64 literal and <= 64 distance trees, one context mode
// for all literal block types, every distance symbol fits its payload form, input < 2^28 bytes, pos + MLEN <= capacity.
// tools/asm_emu.py runs this file instruction by instruction against LDS states dumped on the GPU and the oracle.

// -DLDS_GROW=bytes: the wider instances of the kernel (brx_kernels_l1/l2/l3.hip, BRX_LDS_GROW of brx_device.h) -- that much
// more table memory, everything behind it that much further up
#ifndef LDS_GROW
#define LDS_GROW 0
#endif
#define RMASK 2047
#define RING 2048
#ifdef LDS_TM_LAST
// level 4 (150 KiB of LDS): the table memory LAST -- the fixed areas stay within a DS instruction's 16-bit offset
#define LDS_ITAB 2048
#define LDS_SPARE 2304
#define LDS_CMH 2560
#define LDS_ST 2816
#define LDS_MBW 3008
#define LDS_PAD0 3200
#define LDS_TM 3328
#else
#define LDS_TM 2048
#define LDS_ST (9728+LDS_GROW)
#define LDS_MBW (9920+LDS_GROW)
// the code-length scratch area (Lds::lens, 768 B) is free during the command loop
#define LDS_ITAB (8960+LDS_GROW)   // byte -> context info (filled by prepare_fast_tables)
#define LDS_SPARE (9216+LDS_GROW)  // 12 x 4 B: the two-entry symbol lists of resident one-symbol literal trees
#define LDS_CMH (9472+LDS_GROW)    // context id * 4 -> tree descriptor of the current literal block type (filled at entry)
#define LDS_PAD0 (10112+LDS_GROW)
#endif
#define SYMOFF 68       // symbol list of a tree: after its 17 header words (brx_kernels.hip, "Table layout in table memory")
#define INFOOFF 64      // the header's info word: kind | max_len << 8 | x << 16
// EXEC inside the loop: lanes 0..16 only (the 16 comparator lanes of a lookup, + 1).  Everything uniform needs one lane; copies,
// flushes and input staging set their own mask.  Against all 64 lanes: -6.5 % at 4096 streams (round 2; why is not settled: the
// clock is 2.395 GHz under the headline load either way, and 32 lanes cost the same as 17 -- round 4, profiles/EXPERIMENTS.md).
#define XLOOP 0x1ffff

// ---- SGPRs (s36-s38: scratch during entry)
#define NPOST s4
#define MA s5
#define MB s6
#define SB s7
#define PFREE s8
#define PBASE s9
#define DTREE s10
#define LINKD s[24:25]
#define XFP s[26:27]
#define LITSYM s28
#define WL s39
#define MAXA s40
#define WLSTOP s41
#define INP s[42:43]
#define WENDM1 s44
#define WSAFE s45
// The loop has no end-of-input test: it is poisoned when the refill pulls in dword WSAFE - 1 = (bitend >> 5) - END_MARGIN - 1
// (resumable decode; batches: (bitend >> 5) + 3 -- mbw[MBW_WSAFE], set by the dispatcher),
// the cursor then at most at bitend - 32 * END_MARGIN - 32, and leaves at the next insert&copy symbol or literal run.  Until
// then it takes at most 24 + 24 (insert / copy extra bits) + 15 + 15 + 24 (a distance block switch: type, count, its extra
// bits) + 15 + 24 (distance symbol, extra bits) = 141 bits: four dwords would do, five are kept (eight until round 4: a
// stream's last 36 bytes went through the C++ loop, a third of what a 400-byte stream takes).
#ifndef END_MARGIN
#define END_MARGIN 5
#endif
#define CBASE s46
#define DIST s47
#define RSRC s[48:51]
#define RSRC2 s[16:19]           // the host-visible mirror of the output slot (Dec::mirror), or a resource without records
#define POS s52
#define DSPEC s57               // bit 16 + 2k: distance context k has a one-symbol tree; bit 24 (implicit distance 0): always;
                                // bit 2k: context k does not take the direct lookup (k < 3, or its tree is special)
#define SKEW s53
#define VFL s54
#define FLUSHAT s55
#define WINDOW s56
#define INS s92                 // INS, CPY, DCTX and s95 are one insert&copy record (s_load_dwordx4)
#define CPY s93
#define MBLEFT s61              // scratch: bytes left in the meta-block = MBEND - POS, computed where needed
#define IZ s64
#define LBLEN s65
#define IBLEN s66
#define DBLEN s67
#define PRIOW s60                // wave slot in the SIMD
#define RUN s68                 // literals left in the current run, one behind
#define NXT s[58:59]            // the next dword of the staged input (lane WL of chunk A), zero-extended
#define NXTLO s58
#define NXTHI s59
#define WIN s[62:63]            // the bit window
#define WINLO s62
#define WINHI s63
#define MBEND s69
#define SNAV s70
#define DCTX s94
#define HISYM s71
#define MA2 s71                  // (after the entry code) MA >> 2
#define BFEB s28                 // (after the entry code) s_bfe operand of a literal entry's share as p2
#define CMDW s72
#define EXITC s73
#define IACTAB s[74:75]
#define DICTP s[76:77]
#define LBLEN_REAL s78
#define IBLEN_REAL s79
#define FLAGS s80
#define DCODE s81
#define LINKA s[82:83]
#define T0 s84
#define T1 s85
#define T01 s[84:85]
#define T2 s86
#define T3 s87
#define T4 s88
#define T5 s89
#define T6 s90
#define T7 s91
#define CLEN s96
// -DBRX_SLOTS (the sparse-launch build only: it reads the window from SGPRs in every lane): the literal loop works on a CACHE
// OF FOUR TREES in one register pair -- lanes 16 s .. 16 s + 15 = the tree in slot s -- and is software-pipelined over it: one
// compare + one fetch decode the next literal under ALL FOUR candidate trees while the previous literal's entry is still on its
// way back from LDS; once that entry is there its context picks the slot.  The LDS round trip leaves the per-literal chain.
#if defined(BRX_WIN_SGPR) && !defined(BRX_PROF) && !defined(BRX_NO_SLOTS)
#define BRX_SLOTS
#endif
// Meta-blocks with more than SLOT_OVER + 1 literal trees take the tree cache; up to twelve trees fit the registers of the resident
// loop, which is the faster one on text (few literals per run: profiles/r05_slots_ab.txt).
#ifndef SLOT_OVER
#define SLOT_OVER 11
#endif
#ifndef BRX_PROF
#define LITJ s[20:21]           // where an insert's literals go (.Lhave_lits): the loop of the meta-block's literal mode
#define LITJLO s20
#define LITJHI s21
#ifdef BRX_SLOTS
#define BFEBI s29               // BFEB for a bare context-info byte (offset - 8)
#define VCCB s[22:23]           // compare masks of the two literals in flight (SL_COMPARE)
#define VCCA s[30:31]
#define SLOTT s11               // byte s = index of the literal tree in slot s, 0xff = none
                                // (FLAGS bits 17:16: the slot the next tree goes to, round robin)
#else
#define BFEBI s22               // BFEB for a bare context-info byte (offset - 8)
#endif
#endif
#define LINKB s[98:99]
#define LINKC s[100:101]
// FLAGS bits (17:16: BRX_SLOTS, the cache's next slot): 0 = block counters poisoned (near the end of the input), 3 = one literal tree, resident in VLITL / VLITB, 6 = more than 8 literal trees: the tree cache (BRX_SLOTS),
// 4 = the literal block types differ in context mode (literal entries are plain bytes), 5 = <= RESIDENT_TREES literal trees, resident
// in v70..v93
// ---- VGPRs (v40-v47 are callee-saved in the AMDGPU calling convention: using them would make the wrapper spill them)
#define VZERO v0
#define VLANE v1
#define VLANE4 v2
#define VLANE16 v3
#define VCHA v4
#define VCHB v5
#define VSH v6
#define VS v7
#define VLEN v54
#define VDP v55
#define VRING v65               // lanes 0..3 = the four last distances, most recent first (pushed by one DPP row shift)
#define VPA v8
#define VLHOFF v9
#define VDHOFF v10
#define VB4 v11
#define VPEND v12
#define VITAB v23               // lane l: context info of bytes 4l .. 4l+3 under the current context mode (= LDS_ITAB)
#define VDICTINFO v13
#define VQ v[14:17]
#define VT0 v18
#define VT1 v19
#define VT2 v20
#define VT3 v21
#define VT4 v22
#define VR v24
#define VU v25
#define VLB v[26:27]
#define VLIM v26
#define VBASE v27
#define VE v28
#define VINFO v29
#define VC v30
#define VH v31
#define VWIN v[32:33]
#define VWINLO v32
#define VWINHI v33
#define VI v35
#define VRF v[36:37]
#define VRFLO v36
#define VRFHI v37
#define VLIT v[38:39]
#define VLITL v38
#define VLITB v39
#define VDH v[48:49]
#define VDHV v48
#define VDHB v49
#define VX v50
#define VN v51
#define VHH v52
#define VEX v53
#define VDH4 v64
#define VIAC v[66:67]
#define VIACL v66
#define VIACB v67
#define VTREES v70          // v70 .. v93: limits / bases of up to RESIDENT_TREES (12) resident literal trees (pairs; indexed through M0).
#define VTREES1 v71         // Eight until round 5: text at quality 11 has 9 - 11 trees from ~64 KB on (tests/golden/enc: 9 of 18 fixtures)
#define RESIDENT_TREES 12
#define VCMIDX v55          // (entry only) 4 * tree index per context id
#define VCMAP v103          // lane c: 2 * tree index of context id c (the M0 value of its pair)
// BRX_DIST_RESIDENT: limits / folded bases of the four distance-context trees of the current block type in v104..v111 (pairs,
// reached through M0 like the literal trees): no LDS round trip for a tree's header in front of a distance symbol, at the price of
// three scalar instructions per distance symbol.  Sparse launches -5 % (config 5 48.3 -> 45.9 ms), a full chip -0.7 %
// (profiles/r03_ab.txt; each build at its own best position).  The serial-fetch A/B build (BRX_NO_SPEC) keeps the LDS form.
#define VDTREES v104
#define VDTREES1 v105
#define VD3L v110               // (the pair of distance context 3 by name)
#define VD3B v111
#ifdef BRX_SLOTS
#define VQL v96                 // the four cached literal trees: limits (as counts) ...
#define VQB v97                 // ... and folded bases, lanes 16 s .. 16 s + 15 = slot s
#define VSLOT v98               // lane c: 16 * slot of the tree of context id c, 0x80 = that tree is not cached
#define VSA v99                 // the candidates' entries of the two literals in flight
#define VSB v100
#define VLITS v101              // the entries of a run's literals (lane = RUN at the time), stored to the ring at the run's end
#define VNLANE v102             // ~lane
#endif
#ifndef BRX_NO_SPEC
#define BRX_DIST_RESIDENT
#endif

// The bit window (a VGPR pair with the same value in every lane, or an SGPR pair: see TAKE); bits are taken from its low end; SNAV =
// number of valid bits (>= 32 after a REFILL_CHECK).  A lone wave pays ~4.2 cycles per instruction, ~10 per scalar
// conditional branch that falls through, ~21-25 per taken branch, ~24 for a v_cmp + s_cbranch_vccnz pair, ~50 per LDS /
// scalar-cache round trip, ~16 when a SALU instruction consumes an SGPR a VALU instruction wrote (tools/ubench/issue.hip).
// Sixteen waves on a CU additionally share its scalar ALU (1 instruction per cycle for all of them), the port for
// SGPR-writing VALU instructions (v_cmp, v_readlane: ~1 per cycle) and the branch unit (~0.6 per cycle)
// (tools/ubench/power.hip): on a full chip the scalar instruction count of a command is what bounds it.
// Refill discipline: >= 32 valid bits at .Lcmd; an insert&copy symbol (<= 15) leaves >= 17, enough for a literal or a
// distance symbol (<= 15); every literal, every extra-bit field > 0 and the distance symbol are followed by a check.
// (bring-up, -DBRX_PROF: cycles spent waiting for copies in flight, and how often, go to Lds::pad[10..13])
#ifdef BRX_PROF
#define LDS_PAD LDS_PAD0
.macro PROF_WAIT_VM
    s_waitcnt lgkmcnt(0)
    s_memtime s[16:17]
    s_waitcnt lgkmcnt(0)
    s_waitcnt vmcnt(0)
    s_memtime s[18:19]
    s_waitcnt lgkmcnt(0)
    s_sub_u32 s18, s18, s16
    s_add_u32 s20, s20, s18
    s_add_u32 s21, s21, 1
    s_add_u32 s22, s22, PFREE
.endm
// section timers: s[16:17] = start of the running section; \acc = accumulator of the section that ends here
.macro PROF_MARK acc
    s_memtime s[18:19]
    s_waitcnt lgkmcnt(0)
    s_sub_u32 s17, s18, s16
    s_add_u32 \acc, \acc, s17
    s_mov_b32 s16, s18
.endm
#else
.macro PROF_WAIT_VM
.endm
.macro PROF_MARK acc
.endm
#endif
// Two builds of this file (tools/ubench/power.hip has the measurements behind the choice): a CU has ONE scalar ALU for
// all of its waves (1 instruction per cycle), while its four SIMDs retire two uniform VALU instructions per cycle
// between them.  With 16 streams on a CU the scalar ALU is the busiest unit, so the default build keeps the window
// arithmetic on the vector side; with few waves per CU nothing is contended and the shortest dependent chain wins:
// -DBRX_WIN_SGPR keeps the window in SGPRs (no VGPR -> SGPR hand-overs for the extra-bit fields, a shorter refill).
#ifdef BRX_WIN_SGPR
#define WSRC WINLO
.macro TAKE n, rid
    s_lshr_b64 WIN, WIN, \n
    s_sub_u32 SNAV, SNAV, \n
    s_cbranch_scc1 .Lrf_stub_\rid
.Lrf_back_\rid:
.endm
// \n (an SGPR, <= 24; >= 32 valid bits) extra bits: \dst = \base + (bits << \shift)
.macro TAKE_EXTRA dst, base, n, rid, shift=0
    s_bfm_b32 T0, \n, 0
    s_and_b32 T0, WINLO, T0
    .ifnc \shift,0
    s_lshl_b32 T0, T0, \shift
    .endif
    s_add_u32 \dst, \base, T0
    TAKE \n, \rid
.endm
// the next dword of the staged input enters the window (it was fetched from its lane when the previous one went in, so
// the VALU -> SALU hand-over is long done); leaves the WL == WLSTOP test in SCC
.macro REFILL_CORE
    s_add_u32 SNAV, SNAV, 32                            // (= the number of valid bits before this dword)
    s_lshl_b64 T01, NXT, SNAV
    s_or_b64 WIN, WIN, T01
    s_add_u32 WL, WL, 1
    v_readlane_b32 NXTLO, VCHA, WL
    s_cmp_lg_u32 WL, WLSTOP
.endm
.macro WIN_INIT lo, hi                                  // at entry: the first two dwords, already shifted
    s_mov_b32 WINLO, \lo
    s_mov_b32 WINHI, \hi
    v_readlane_b32 NXTLO, VCHA, 2
    s_mov_b32 NXTHI, 0
.endm
.macro WIN_ROLLED                                       // after the input staging rolled (WL = 0)
    v_readlane_b32 NXTLO, VCHA, 0
.endm
#else
#define WSRC VWINLO
.macro TAKE n, rid
    v_lshrrev_b64 VWIN, \n, VWIN
    s_sub_u32 SNAV, SNAV, \n                             // borrow = fewer than 32 valid bits left
    s_cbranch_scc1 .Lrf_stub_\rid
.Lrf_back_\rid:
.endm
.macro TAKE_EXTRA dst, base, n, rid, mul=0                // (\mul: an SGPR with the factor of the extra bits, \base then a VGPR)
    v_bfe_u32 VEX, VWINLO, 0, \n
    .ifnc \mul,0
    v_mad_u32_u24 VEX, VEX, \mul, \base                 // (at most 24 extra bits, a factor of at most 8)
    .else
    v_add_u32 VEX, \base, VEX
    .endif
    TAKE \n, \rid
    v_readfirstlane_b32 \dst, VEX
.endm
.macro REFILL_CORE
    v_readlane_b32 T0, VCHA, WL
    v_mov_b32 VRFHI, 0
    s_nop 1                                             // gfx940+: VALU-written SGPR read by a VALU: 2 wait states
    v_mov_b32 VRFLO, T0
    s_add_u32 SNAV, SNAV, 32                            // (= the number of valid bits before this dword)
    v_lshlrev_b64 VRF, SNAV, VRF
    v_or_b32 VWINLO, VWINLO, VRFLO
    v_or_b32 VWINHI, VWINHI, VRFHI
    s_add_u32 WL, WL, 1
    s_cmp_lg_u32 WL, WLSTOP
.endm
.macro WIN_INIT lo, hi
    v_mov_b32 VWINLO, \lo
    v_mov_b32 VWINHI, \hi
.endm
.macro WIN_ROLLED
.endm
#endif
// Out-of-line part of a refill: next dword of the staged input (lane WL of chunk A) enters the window.
.macro REFILL_STUB id
.Lrf_stub_\id:
    REFILL_CORE
    s_cbranch_scc1 .Lrf_back_\id
    s_call_b64 LINKA, .Lspecial
    s_branch .Lrf_back_\id
.endm
// A register-resident tree keeps its limits as COUNTS -- lane L: limit[L] in units of L-bit codes (first_code + count), the
// header word >> (31 - L) -- so that its compare works on the candidate index the lane computes anyway (the top L bits of
// the window) and no 31-bit copy of the window is needed: one VALU instruction less per lookup (profiles/r03_ab.txt).
// (Lane 0 of a general tree stays 0 = never; the all-ones lane 0 of a resident one-symbol tree stays "always".)
.macro TO_COUNTS reg, tmp
    v_add_u32 \tmp, -1, VSH
    v_lshrrev_b32 \reg, \tmp, \reg
.endm
// A tree's header words as loaded (lane L: limit[L] << 16 | base[L] & 0xffff, lane 0: 0) -> the pair the lookups work on:
// \lim = limit[L] << 16, \base = base[L] sign-extended.
.macro SPLIT_TREE lim, base
    v_bfe_i32 \base, \lim, 0, 16
    v_and_b32 \lim, 0xffff0000, \lim
.endm
// Canonical prefix-code lookup (table layout: brx_kernels.hip, "Table layout in table memory").
// lim = per-lane limit[L] << 16 (lane 0: 0), base = per-lane base[L] of the tree (lane L, L = 1..15; lanes >= 16 repeat).
// Out: CLEN = code length (SGPR), VI = index into the tree's sorted symbol list (VGPR).  Clobbers T2, T3, VR, VU, vcc.
// Every code is complete (precondition), so some lane always matches; the lowest matching lane is the length.
.macro LOOKUP lim, base
    v_bfrev_b32 VR, WSRC
    v_lshrrev_b32 VU, 1, VR
    v_cmp_lt_u32 vcc, VU, \lim
    s_ff1_i32_b32 CLEN, vcc_lo
    v_readlane_b32 T3, \base, CLEN
    s_sub_u32 T2, 32, CLEN
    v_lshrrev_b32 VI, T2, VR                            // (two instructions between the v_readlane and its VALU reader)
    v_add_u32 VI, T3, VI
.endm

// The same lookup with the symbol fetch taken off the dependent chain: every lane L computes the list index its length
// would give (per-lane shift VSH = 32 - L) and fetches that entry speculatively; the code length (s_ff1 of the compare)
// then only selects the lane (v_readlane \rd-result, CLEN) after the LDS round trip.  Candidates of the wrong lengths
// read harmless addresses (out-of-range LDS reads return 0).  Consumes the code's bits.  Out: CLEN, VS (lane CLEN = the entry).
#ifdef BRX_NO_SPEC
// (A/B switch, BRX_NO_SPEC=1 at build time: the same lookup with the fetch BEHIND the length -- ballot, s_ff1, base of that
// length, then one fetch of the one entry; every lane ends up with the same entry.  profiles/r02_spec_ab.txt)
.macro LOOKUP2 lim, base, symbase, scale, rd, off, rid
    v_bfrev_b32 VR, WSRC
    v_lshrrev_b32 VU, 1, VR
    v_cmp_lt_u32 vcc, VU, \lim
    s_ff1_i32_b32 CLEN, vcc_lo
    v_readlane_b32 T3, \base, CLEN
    s_sub_u32 T2, 32, CLEN
    s_min_u32 T2, T2, 31                                // ("length 0" of a resident one-symbol tree: index 0 or 1)
    v_lshrrev_b32 VI, T2, VR
    v_add_u32 VI, T3, VI
    v_lshl_add_u32 VI, VI, \scale, \symbase
    \rd VS, VI offset:\off
    TAKE CLEN, \rid
.endm
.macro LOOKUP2F lim, basep, scale, rd, rid
    v_bfrev_b32 VR, WSRC
    v_lshrrev_b32 VU, VSH, VR                           // (limits of a resident tree are counts: TO_COUNTS)
    v_cmp_lt_u32 vcc, VU, \lim
    s_ff1_i32_b32 CLEN, vcc_lo
    v_readlane_b32 T3, \basep, CLEN
    s_sub_u32 T2, 32, CLEN
    s_min_u32 T2, T2, 31
    v_lshrrev_b32 VI, T2, VR
    v_lshl_add_u32 VI, VI, \scale, T3
    \rd VS, VI
    TAKE CLEN, \rid
.endm
.macro LOOKUP2X rid
    v_bfrev_b32 VR, WSRC
    v_lshrrev_b32 VU, VSH, VR
    s_set_gpr_idx_on T6, 2
    v_cmp_lt_u32 vcc, VU, VTREES
    s_set_gpr_idx_off
    s_ff1_i32_b32 CLEN, vcc_lo
    s_set_gpr_idx_on T6, 1
    v_readlane_b32 T3, VTREES1, CLEN
    s_set_gpr_idx_off
    s_sub_u32 T2, 32, CLEN
    s_min_u32 T2, T2, 31
    v_lshrrev_b32 VI, T2, VR
    v_lshl_add_u32 VI, VI, 1, T3
    ds_read_u16 VS, VI
    TAKE CLEN, \rid
.endm
#else
.macro LOOKUP2 lim, base, symbase, scale, rd, off, rid
    v_bfrev_b32 VR, WSRC
    v_lshrrev_b32 VU, 1, VR
    v_cmp_lt_u32 vcc, VU, \lim
    v_lshrrev_b32 VI, VSH, VR
    v_add_u32 VI, \base, VI
    v_lshl_add_u32 VI, VI, \scale, \symbase
    \rd VS, VI offset:\off
    s_ff1_i32_b32 CLEN, vcc_lo                          // code length
    TAKE CLEN, \rid
.endm
// The lookup in one of the register-resident literal trees, T6 = 2 * its index: the VGPR index mode (gfx9 has no
// v_movrel) redirects the second source of the compare (limits) and the third of the shift-add (folded bases).
.macro LOOKUP2X rid
    v_bfrev_b32 VR, WSRC
    v_lshrrev_b32 VI, VSH, VR
    s_set_gpr_idx_on T6, 6                              // SRC1 | SRC2 + T6
    v_cmp_lt_u32 vcc, VI, VTREES
    v_lshl_add_u32 VI, VI, 1, VTREES1
    s_set_gpr_idx_off
    ds_read_u16 VS, VI
    s_ff1_i32_b32 CLEN, vcc_lo
    TAKE CLEN, \rid
.endm
// The lookup of a tree that lives in registers: \basep = per-lane (base[L] << scale) + LDS address of the symbol list,
// folded once when the tree is loaded, so the candidate address is one shift and one shift-add.
.macro LOOKUP2F lim, basep, scale, rd, rid
    v_bfrev_b32 VR, WSRC
    v_lshrrev_b32 VI, VSH, VR
    v_cmp_lt_u32 vcc, VI, \lim                          // (limits as counts: TO_COUNTS)
    v_lshl_add_u32 VI, VI, \scale, \basep
    \rd VS, VI
    s_ff1_i32_b32 CLEN, vcc_lo                          // code length
    TAKE CLEN, \rid
.endm
// The same in two halves: the compare and the speculative fetches are issued EARLY (at .Lcopy: the next insert&copy symbol's
// bits are known as soon as the distance is), the length and the TAKE follow at .Lcmd_pre -- the fetch's LDS round trip runs
// under the copy's bookkeeping.  vcc, VS (and VR, VU, VI) must survive in between: the common copy paths write none of them.
.macro LOOKUP2F_EARLY lim, basep, scale, rd
    v_bfrev_b32 VR, WSRC
    v_lshrrev_b32 VI, VSH, VR
    v_cmp_lt_u32 vcc, VI, \lim
    v_lshl_add_u32 VI, VI, \scale, \basep
    \rd VS, VI
.endm
.macro LOOKUP2F_LATE rid
    s_ff1_i32_b32 CLEN, vcc_lo                          // code length
    TAKE CLEN, \rid
.endm
#endif

// Descriptors (LDS byte address of the header, or 0x80000000 | x for a one-symbol tree) of the trees whose INDICES sit in the lanes of
// \vidx -> \vd (may be the same register), gathered from table memory: tm[list + index] = h, the header's info word.  \mbwoff = where
// the meta-block words hold the list's word address (20: literal trees, 28: distance trees).  What the lanes of VLHOFF / VDHOFF give
// for the first 64 trees; a meta-block with more trees of a kind (one piece of > 1 MiB from libbrotlienc: up to 256 literal trees,
// profiles/r05_big_trees.txt) takes this on entry and at block switches.  EXEC = the lanes wanted; clobbers \vta, \vtb, \mask.
.macro DESC_GATHER vd, vidx, mbwoff, kmask, vta, vtb, mask
    ds_read_b32 \vta, VZERO offset:LDS_MBW+\mbwoff
    s_waitcnt lgkmcnt(0)
    v_add_u32 \vd, \vidx, \vta
    v_lshlrev_b32 \vd, 2, \vd
    ds_read_b32 \vd, \vd offset:LDS_TM                  // h
    s_waitcnt lgkmcnt(0)
    v_lshlrev_b32 \vd, 2, \vd
    ds_read_b32 \vta, \vd offset:LDS_TM+INFOOFF         // kind | max_len << 8 | x << 16
    v_add_u32 \vd, LDS_TM, \vd
    s_waitcnt lgkmcnt(0)
    v_and_b32 \vtb, \kmask, \vta
    v_lshrrev_b32 \vta, 16, \vta
    v_or_b32 \vta, 0x80000000, \vta
    v_cmp_eq_u32 \mask, 1, \vtb
    s_nop 1                                             // (a VALU-written SGPR pair as the next VALU's mask)
    v_cndmask_b32 \vd, \vd, \vta, \mask
.endm
// ======================================================================================================== entry
    .p2align 8
    s_getreg_b32 PRIOW, hwreg(HW_REG_HW_ID, 0, 4)       // this wave's slot in its SIMD (phase of the priority rotation, .Lspecial)
    s_waitcnt vmcnt(0) lgkmcnt(0)
    v_mov_b32 VZERO, 0
    v_mbcnt_lo_u32_b32 VLANE, -1, 0
    v_mbcnt_hi_u32_b32 VLANE, -1, VLANE
    v_and_b32 VLANE4, 15, VLANE
    v_lshlrev_b32 VLANE4, 2, VLANE4                     // 4 * (lane & 15): lane L reads header word L of a tree
    v_lshlrev_b32 VLANE16, 4, VLANE
    v_and_b32 VSH, 15, VLANE
    v_sub_u32 VSH, 32, VSH                              // LOOKUP2: lane L shifts the reversed window by 32 - L
    v_min_u32 VSH, 31, VSH                              // (lane 0 = "length 0" of a one-symbol tree: index 0 or 1 of its 2-entry list)
    // parked decoder state: st[0..17], st[23..26], st[36..37]
    ds_read_b128 v[20:23], VZERO offset:LDS_ST+0       // in_words lo, hi, w_end, bitpos lo
    ds_read_b128 v[24:27], VZERO offset:LDS_ST+16      // bitpos hi, bitend lo, bitend hi, out lo
    ds_read_b128 v[28:31], VZERO offset:LDS_ST+32      // out hi, cap, pos, a
    ds_read_b128 v[32:35], VZERO offset:LDS_ST+48      // vfl, window, dist0, dist1
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 s42, v20
    v_readfirstlane_b32 s43, v21
    v_readfirstlane_b32 T0, v22                         // w_end
    v_readfirstlane_b32 T1, v23                         // bitpos
    v_readfirstlane_b32 T2, v25                         // bitend
    v_readfirstlane_b32 s48, v27
    v_readfirstlane_b32 s49, v28
    v_readfirstlane_b32 s50, v29                        // num_records = the stream's capacity: a wild offset is dropped, not a fault
    v_readfirstlane_b32 POS, v30
    v_readfirstlane_b32 SKEW, v31
    v_readfirstlane_b32 VFL, v32
    v_readfirstlane_b32 WINDOW, v33
    v_min_u32 VRING, 3, VLANE
    v_lshlrev_b32 VRING, 2, VRING
    ds_read_b32 VRING, VRING offset:LDS_ST+56          // lane k = dist k
    ds_read_b32 v22, VZERO offset:LDS_ST+92             // t_dict (st[23], st[24]: only 4-byte aligned)
    ds_read_b32 v23, VZERO offset:LDS_ST+96
    ds_read_b32 v26, VZERO offset:LDS_ST+100            // t_xforms (st[25], st[26])
    ds_read_b32 v27, VZERO offset:LDS_ST+104
    ds_read_b32 v24, VZERO offset:LDS_ST+108            // t_lut (st[27], st[28])
    ds_read_b32 v25, VZERO offset:LDS_ST+112
    ds_read_b64 v[28:29], VZERO offset:LDS_ST+144       // insert&copy / dictionary info table
    s_and_b32 s49, s49, 0xffff
    s_mov_b32 s51, 0x00020000
#ifndef BRX_PROF
    ds_read_b64 v[30:31], VZERO offset:LDS_ST+160       // st[40], st[41]: mirror of the output slot in host memory, or 0
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 s16, v30
    v_readfirstlane_b32 s17, v31
    s_and_b32 s17, s17, 0xffff
    s_or_b32 s18, s16, s17
    s_cmp_lg_u32 s18, 0
    s_cselect_b32 s18, s50, 0                           // no mirror: no records, every store to it is dropped
    s_mov_b32 s19, 0x00020000
#endif
    s_sub_u32 WENDM1, T0, 1
    s_lshr_b32 CBASE, T1, 5                             // first staged dword = the one holding the cursor
    s_and_b32 T3, T1, 31                                // bit offset inside it
    // stage 2 x 64 input dwords (clamped to the stream's last dword)
    v_add_u32 VT0, CBASE, VLANE
    v_min_u32 VT0, WENDM1, VT0
    v_lshlrev_b32 VT0, 2, VT0
    global_load_dword VCHA, VT0, INP
    s_add_u32 T4, CBASE, 64
    v_add_u32 VT0, T4, VLANE
    v_min_u32 VT0, WENDM1, VT0
    v_lshlrev_b32 VT0, 2, VT0
    global_load_dword VCHB, VT0, INP
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 s76, v22
    v_readfirstlane_b32 s77, v23
    v_readfirstlane_b32 T4, v24
    v_readfirstlane_b32 T5, v25
    v_readfirstlane_b32 s26, v26
    v_readfirstlane_b32 s27, v27
    v_readfirstlane_b32 s74, v28
    v_readfirstlane_b32 s75, v29
    // context LUTs (3 x 64 dwords) and the dictionary info vector (dwords 2816.. of the insert&copy table): lane n =
    // DOFFSET[n] | NDBITS[n] << 24
    v_lshlrev_b32 VT0, 2, VLANE
    v_add_u32 VT1, 11264, VT0
    s_nop 4                                             // v_readfirstlane -> VMEM address SGPR: 5 wait states
    global_load_dword v28, VT0, s[88:89]                // Lut0
    global_load_dword v29, VT0, s[88:89] offset:256     // Lut1
    global_load_dword v30, VT0, s[88:89] offset:512     // Lut2
    global_load_dword VDICTINFO, VT1, IACTAB
    // meta-block words
    ds_read_b128 v[20:23], VZERO offset:LDS_MBW+0       // npostfix, ndirect, cmode_w, cml
    ds_read_b128 v[24:27], VZERO offset:LDS_MBW+16      // cmd, hl, hi, hd
    ds_read_b64 v[32:33], VZERO offset:LDS_MBW+32       // ntl, ntd
    ds_read_b32 v36, VZERO offset:LDS_MBW+124           // MBW_ASM: 1 | mixed context modes << 1
    ds_read_b32 v37, VZERO offset:LDS_MBW+156           // MBW_WSAFE
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 FLAGS, v36
    // Where the loop gets poisoned (.Lspecial): the dispatcher says (mbw[MBW_WSAFE], round 6).  The resumable decode of the bounded
    // reader, which cannot take anything back: END_MARGIN dwords in front of the end of the input, every bit consumed here is a real
    // bit.  Batches: four dwords BEHIND the end -- a valid stream's last meta-block ends before that (the staged input repeats the
    // stream's last dword beyond its end: memory-safe), so its last bytes do not go through the C++ loop; a truncated stream runs on
    // into those repeated bits, is poisoned and leaves with its cursor beyond the end, and the kernel then goes back to a checkpoint
    // of the parked state that it took a few dozen dwords in front of the end -- which is where the loop is poisoned FIRST in a long
    // stream -- and decodes the rest with the exact rules (brx_kernels.hip, "speculative end").
    v_readfirstlane_b32 WSAFE, v37
    v_readfirstlane_b32 NPOST, v20
#ifndef BRX_WIN_SGPR
    s_lshl_b32 NPOST, 1, NPOST                          // (this build multiplies: TAKE_EXTRA)
#endif
    v_readfirstlane_b32 T0, v22                         // cmode_w
    v_readfirstlane_b32 T1, v23                         // cml
    v_readfirstlane_b32 T2, v24                         // cmd
    v_readfirstlane_b32 T6, v25                         // hl
    v_readfirstlane_b32 T7, v26                         // hi
    v_readfirstlane_b32 s12, v27                        // hd
    v_readfirstlane_b32 s13, v32                        // ntl
    v_readfirstlane_b32 s14, v33                        // ntd
    ds_read_b32 v20, VZERO offset:LDS_MBW+52            // L.btype
    ds_read_b32 v21, VZERO offset:LDS_MBW+60            // L.blen
    ds_read_b32 v22, VZERO offset:LDS_MBW+76            // I.btype
    ds_read_b32 v23, VZERO offset:LDS_MBW+84            // I.blen
    ds_read_b32 v24, VZERO offset:LDS_MBW+100           // D.btype
    ds_read_b32 v25, VZERO offset:LDS_MBW+108           // D.blen
    ds_read_b128 v[32:35], VZERO offset:LDS_MBW+128     // mb_left, insert_len, copy_len, implicit_zero
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 s15, v20                        // L.btype
    v_readfirstlane_b32 LBLEN, v21
    v_readfirstlane_b32 s97, v22                        // I.btype
    v_readfirstlane_b32 IBLEN, v23
    v_readfirstlane_b32 DCODE, v24                      // D.btype (DCODE is free until the first distance)
    v_readfirstlane_b32 DBLEN, v25
    v_readfirstlane_b32 MBLEFT, v32
    s_add_u32 MBEND, MBLEFT, POS                        // end of the meta-block
    v_readfirstlane_b32 INS, v33
    v_readfirstlane_b32 CPY, v34
    v_readfirstlane_b32 IZ, v35
    // per-tree descriptors: lane t -> LDS byte address of tree t's header (tm word h -> LDS_TM + 4h), or for a
    // one-symbol tree (kind 1, zero-bit code, SURVEY Q5) 0x80000000 | x  (literal: x = byte | info << 8)
    s_and_b32 FLAGS, FLAGS, 2
    s_lshl_b32 FLAGS, FLAGS, 3                          // bit 4: mixed context modes
    s_sub_u32 s13, s13, 1
    v_min_u32 VT0, s13, VLANE
    v_add_u32 VT0, T6, VT0
    v_lshlrev_b32 VT0, 2, VT0
    ds_read_b32 VLHOFF, VT0 offset:LDS_TM
    s_sub_u32 s14, s14, 1
    v_min_u32 VT0, s14, VLANE
    v_add_u32 VT0, s12, VT0
    v_lshlrev_b32 VT0, 2, VT0
    ds_read_b32 VDHOFF, VT0 offset:LDS_TM
    s_waitcnt lgkmcnt(0)
    v_lshlrev_b32 VLHOFF, 2, VLHOFF
    v_lshlrev_b32 VDHOFF, 2, VDHOFF
    ds_read_b32 VT0, VLHOFF offset:LDS_TM+INFOOFF       // the header's info word: kind | max_len << 8 | x << 16
    ds_read_b32 VT1, VDHOFF offset:LDS_TM+INFOOFF
    v_add_u32 VLHOFF, LDS_TM, VLHOFF
    v_add_u32 VDHOFF, LDS_TM, VDHOFF
    s_waitcnt lgkmcnt(0)
    v_and_b32 VT2, 3, VT0
    v_lshrrev_b32 VT0, 16, VT0
    v_or_b32 VT0, 0x80000000, VT0
    v_cmp_eq_u32 vcc, 1, VT2
    v_cndmask_b32 VLHOFF, VLHOFF, VT0, vcc
    v_and_b32 VT2, 7, VT1                               // (kind 5: a one-symbol explicit-distance code made a table, prepare_fast_tables: general here)
    v_lshrrev_b32 VT1, 16, VT1
    v_or_b32 VT1, 0x80000000, VT1
    v_cmp_eq_u32 vcc, 1, VT2
    v_cndmask_b32 VDHOFF, VDHOFF, VT1, vcc
    // insert&copy tree of the current block type: its limits and bases stay resident
    s_add_u32 T7, T7, s97
    s_lshl_b32 T7, T7, 2
    v_mov_b32 VT0, T7
    ds_read_b32 VT1, VT0 offset:LDS_TM
    // literal context map row of the current block type (64 bytes), distance map word, context mode
    s_lshl_b32 T6, s15, 6
    s_add_u32 T1, T1, T6
    v_add_u32 VT0, T1, VLANE
    ds_read_u8 VT4, VT0 offset:LDS_TM                   // lane c: tree index of context id c
    s_lshl_b32 T6, DCODE, 2
    s_add_u32 T2, T2, T6
    v_mov_b32 VT0, T2
    ds_read_b32 VT2, VT0 offset:LDS_TM
    s_lshl_b32 T0, T0, 2
    s_add_u32 T0, T0, s15
    v_mov_b32 VT0, T0
    ds_read_u8 VT3, VT0 offset:LDS_TM
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 T7, VT1                         // h of the insert&copy tree
    v_readfirstlane_b32 CMDW, VT2
    v_readfirstlane_b32 T0, VT3                         // context mode
    // CMH[c] = descriptor of the literal tree of context id c; VDH4 lane 2k = descriptor of the distance tree of
    // distance context k (both for the current block types; a block switch leaves the loop and re-enters here)
    v_lshlrev_b32 VCMIDX, 2, VT4
    s_cmp_gt_u32 s13, 63
    s_cbranch_scc1 .Lent_cmh_far                        // more than 64 literal trees: FLAGS bit 7, DESC_GATHER
    ds_bpermute_b32 VT4, VCMIDX, VLHOFF
.Lent_cmh_have:
    v_lshrrev_b32 VT3, 1, VLANE                         // lane 2k (and 2k + 1) = context k
    v_lshlrev_b32 VT3, 3, VT3
    v_lshrrev_b32 VT3, VT3, CMDW
    v_and_b32 VT3, 0xff, VT3
    s_cmp_gt_u32 s14, 63
    s_cbranch_scc1 .Lent_vdh_far                        // more than 64 distance trees: FLAGS bit 8
    v_lshlrev_b32 VT3, 2, VT3
    ds_bpermute_b32 VDH4, VT3, VDHOFF
.Lent_vdh_have:
    v_lshlrev_b32 VT3, 2, VLANE
    s_waitcnt lgkmcnt(0)
    ds_write_b32 VT3, VT4 offset:LDS_CMH
    s_lshl_b32 T7, T7, 2
    s_add_u32 T7, T7, LDS_TM
    s_add_u32 HISYM, T7, SYMOFF
    v_add_u32 VT0, T7, VLANE4
    ds_read_b32 VIACL, VT0                              // header words of the insert&copy tree (split below, SPLIT_TREE)
    // context masks of the mode (context_info() in brx_kernels.hip): id * 4 = (info(p1) & MA) | ((info(p2) & MB) << SB)
    s_mov_b32 MA, 0xfc
    s_mov_b32 MB, 0
    s_mov_b32 SB, 0
    s_cmp_eq_u32 T0, 2
    s_cselect_b32 MB, 3, MB
    s_cselect_b32 SB, 2, SB
    s_cmp_eq_u32 T0, 3
    s_cselect_b32 MA, 0xe0, MA
    s_cselect_b32 MB, 0x1c, MB
    // byte -> context info of the current block type's mode, 4 bytes per lane (context_info() in brx_kernels.hip)
    s_waitcnt vmcnt(0)
    s_mov_b32 T6, 0x01010101
    v_lshlrev_b32 VT0, 2, VLANE
    v_and_b32 VT0, 0x3f, VT0
    v_mul_lo_u32 VT0, VT0, T6
    v_add_u32 VT1, 0x03020100, VT0                      // mode 0: p & 63 of bytes 4l .. 4l+3
    s_cmp_eq_u32 T0, 1
    s_cbranch_scc0 .Lit_not1
    v_mul_lo_u32 VT1, VLANE, T6                         // mode 1: p >> 2 = l
.Lit_not1:
    v_and_b32 VT1, 0x3f3f3f3f, VT1
    v_lshlrev_b32 VT1, 2, VT1
    s_cmp_eq_u32 T0, 2
    s_cbranch_scc0 .Lit_not2
    v_and_b32 VT1, 0x3f3f3f3f, v28                      // mode 2: Lut0 << 2 | Lut1
    v_lshlrev_b32 VT1, 2, VT1
    v_and_b32 VT2, 0x03030303, v29
    v_or_b32 VT1, VT1, VT2
.Lit_not2:
    s_cmp_eq_u32 T0, 3
    s_cbranch_scc0 .Lit_not3
    v_and_b32 VT2, 0x07070707, v30                      // mode 3: Lut2 << 5 | Lut2 << 2
    v_lshlrev_b32 VT1, 5, VT2
    v_lshlrev_b32 VT2, 2, VT2
    v_or_b32 VT1, VT1, VT2
.Lit_not3:
    v_lshlrev_b32 VT0, 2, VLANE
    ds_write_b32 VT0, VT1 offset:LDS_ITAB
    v_mov_b32 VITAB, VT1                                // (resident copy: LIT_CTX_ENTRY)
    // up to 8 literal trees, one context mode: limits and bases live in v70..v85 (pair t = tree t, reached through M0),
    // lane 16 of a tree's limits carries the LDS address of its symbol list.  A one-symbol tree becomes a real table:
    // limit[0] = all ones ("matches" at length 0, no bits), a two-entry list [x, x].
#ifdef BRX_SLOTS
    v_not_b32 VNLANE, VLANE
#endif
    s_bitcmp1_b32 FLAGS, 4
    s_cbranch_scc1 .Lent_no_r
#ifdef BRX_SLOTS
    s_cmp_gt_u32 s13, SLOT_OVER
    s_cbranch_scc1 .Lent_slots
#endif
    s_cmp_gt_u32 s13, RESIDENT_TREES - 1
    s_cbranch_scc1 .Lent_no_r
    s_bitset1_b32 FLAGS, 5
    v_lshrrev_b32 VCMAP, 1, VCMIDX                      // (VCMIDX = 4 * tree index per context id)
    s_mov_b32 T6, 0
.Lent_r_loop:
    v_readlane_b32 T7, VLHOFF, T6                       // descriptor of tree T6
    s_cmp_lt_i32 T7, 0
    s_cbranch_scc1 .Lent_r_single
    v_add_u32 VT0, T7, VLANE4
    ds_read_b32 VLIM, VT0
    s_add_u32 T7, T7, SYMOFF
    s_waitcnt lgkmcnt(0)
    SPLIT_TREE VLIM, VBASE
    v_lshl_add_u32 VBASE, VBASE, 1, T7                  // folded bases (LOOKUP2F)
    TO_COUNTS VLIM, VT1
    s_branch .Lent_r_store
.Lent_r_single:
    v_mov_b32 VLIM, 0
    s_mov_b32 T5, -1
    v_writelane_b32 VLIM, T5, 0
    s_and_b32 T5, T7, 0xffff                            // x = byte | context info << 8
    s_lshl_b32 T4, T5, 16
    s_or_b32 T5, T5, T4
    s_lshl_b32 T4, T6, 2
    s_add_u32 T4, T4, LDS_SPARE
    v_mov_b32 VT0, T4
    v_mov_b32 VT1, T5
    ds_write_b32 VT0, VT1
    v_mov_b32 VBASE, T4
.Lent_r_store:
    s_lshl_b32 T4, T6, 1
    s_set_gpr_idx_on T4, 8                              // VGPR index mode, destination + T4 (gfx9 has no v_movrel)
    v_mov_b32 VTREES, VLIM
    v_mov_b32 VTREES1, VBASE
    s_set_gpr_idx_off
    s_add_u32 T6, T6, 1
    s_cmp_le_u32 T6, s13
    s_cbranch_scc1 .Lent_r_loop
#ifdef BRX_SLOTS
    s_branch .Lent_no_r
    // more than RESIDENT_TREES literal trees (one context mode): the tree cache of the pipelined literal loop -- trees 0 .. 3 go in at once,
    // the others when a literal first needs them (.Lslot_fill)
.Lent_slots:
    s_bitset1_b32 FLAGS, 6
    v_lshrrev_b32 VCMAP, 1, VCMIDX                      // (VCMIDX = 4 * tree index per context id)
    s_mov_b32 SLOTT, -1
    v_mov_b32 VSLOT, 0x80
    v_mov_b32 VQL, 0
    v_mov_b32 VQB, 0
    s_mov_b32 T7, 0
.Lent_s_loop:
    s_mov_b32 T6, T7
    s_call_b64 LINKB, .Lslot_fill
    s_add_u32 T7, T7, 1
    s_cmp_lt_u32 T7, 4
    s_cbranch_scc1 .Lent_s_loop
#endif
.Lent_no_r:
    // one literal tree (and a general one): keep it in registers, no contexts
    s_cmp_lg_u32 s13, 0
    s_cbranch_scc1 .Lent_multi
    v_readfirstlane_b32 T6, VLHOFF
    s_cmp_lt_i32 T6, 0
    s_cbranch_scc1 .Lent_multi
    s_bitset1_b32 FLAGS, 3
    s_add_u32 LITSYM, T6, SYMOFF
    v_add_u32 VT0, T6, VLANE4
    ds_read_b32 VLITL, VT0
.Lent_multi:
    // bit window
    s_waitcnt vmcnt(0)
    v_readlane_b32 s36, VCHA, 0
    v_readlane_b32 s37, VCHA, 1
    s_lshr_b64 s[36:37], s[36:37], T3
    s_sub_u32 SNAV, 32, T3                              // SNAV = valid bits - 32 (TAKE)
    WIN_INIT s36, s37
    s_mov_b32 WL, 2
    s_sub_u32 T0, WSAFE, CBASE
    s_cselect_b32 T0, 0, T0
    s_min_u32 WLSTOP, T0, 64
#ifdef BRX_PROF
    s_mov_b32 s20, 0
    s_mov_b32 s21, 0
    s_mov_b32 s22, 0
    s_mov_b32 s23, 0
    s_mov_b32 s29, 0
    s_mov_b32 s30, 0
    s_mov_b32 s31, 0
    s_memtime s[12:13]
    s_memtime s[16:17]
    s_waitcnt lgkmcnt(0)
#endif
    s_mov_b32 PFREE, 0
    s_mov_b32 PBASE, POS
    s_sub_u32 DCTX, CPY, 2                              // distance context of the parked command; 4 = implicit distance 0
    s_min_u32 DCTX, DCTX, 3
    s_cmp_lg_u32 IZ, 0
    s_cselect_b32 DCTX, 4, DCTX
    s_mov_b32 T0, 0xc0000000
    s_lshl_b32 DCTX, DCTX, 1                            // (kept doubled: VDH4 lane and resident register pair in one)
    v_writelane_b32 VDH4, T0, 8                         // "tree" of an implicit distance code 0
#ifdef BRX_DIST_RESIDENT
    s_call_b64 LINKB, .Lload_dtrees
#endif
    s_mov_b32 EXITC, 1
    s_and_b32 T0, VFL, 0xfffffc00
    s_add_u32 T0, T0, 1024                              // BRX_FLUSH_BLOCK + BRX_FLUSH_LAG
    s_sub_u32 FLUSHAT, T0, SKEW
    s_waitcnt lgkmcnt(0)
    // not enough input left for the fast loop, or ragged flush cursor: hand straight back
    s_cmp_lt_u32 WLSTOP, 3
    s_cbranch_scc1 .Lexit
    s_and_b32 T0, VFL, 1023
    s_cmp_lg_u32 T0, 0
    s_cbranch_scc1 .Lexit
    SPLIT_TREE VIACL, VIACB
    SPLIT_TREE VLITL, VLITB
    v_lshl_add_u32 VIACB, VIACB, 1, HISYM               // folded bases of the resident trees (LOOKUP2F)
    v_lshl_add_u32 VLITB, VLITB, 1, LITSYM
    TO_COUNTS VIACL, VT3
    TO_COUNTS VLITL, VT3
    // (HISYM and LITSYM are free from here on) the context masks of the resident literal loop, which works on id, not id * 4:
    // id = ((info >> 2) & MA2) | share of the previous literal; share = ((info & MB) << SB) >> 2 = a bit field of the entry
    s_lshr_b32 MA2, MA, 2
    s_mov_b32 BFEB, 0
    s_cmp_eq_u32 SB, 2
    s_cselect_b32 BFEB, 0x20008, BFEB                   // mode 2: info bits 1:0
    s_cmp_eq_u32 MB, 0x1c
    s_cselect_b32 BFEB, 0x3000a, BFEB                   // mode 3: info bits 4:2
#ifndef BRX_PROF
    s_sub_u32 BFEBI, BFEB, 8
    s_cmp_eq_u32 BFEB, 0
    s_cselect_b32 BFEBI, 0, BFEBI
    // one indirect branch per insert instead of the chain of FLAGS tests
    s_getpc_b64 LITJ
.Llitj_base:
    s_mov_b32 T0, .Llit_entry_g-.Llitj_base
    s_bitcmp1_b32 FLAGS, 5
    s_cselect_b32 T0, .Llit_r_entry_mx-.Llitj_base, T0
    s_cselect_b32 T1, 0x1c, -1                          // (... of context mode 3: MB = 0x1c)
    s_cmp_eq_u32 T1, MB
    s_cselect_b32 T0, .Llit_r_entry_m3-.Llitj_base, T0
#ifdef BRX_SLOTS
    s_bitcmp1_b32 FLAGS, 6
    s_cselect_b32 T0, .Llit_r_entry_sx-.Llitj_base, T0
    s_cselect_b32 T1, 0x1c, -1
    s_cmp_eq_u32 T1, MB
    s_cselect_b32 T0, .Llit_r_entry_s3-.Llitj_base, T0
#endif
    s_bitcmp1_b32 FLAGS, 3
    s_cselect_b32 T0, .Lhave_lits1-.Llitj_base, T0
    s_add_u32 LITJLO, LITJLO, T0
    s_addc_u32 LITJHI, LITJHI, 0
#endif
    s_mov_b32 MAXA, 0
    s_mov_b64 exec, XLOOP
    s_branch .Lr1

// ======================================================================================================== R0
// Where the loop sits relative to the 32-byte instruction fetch windows is worth 2 - 5 % (measured: all 8 dword offsets, both
// builds; profiles/r02_alignment.txt, profiles/r03_pins.txt): pin it -- 256-byte alignment right in FRONT OF THE LOOP (the entry
// code above ends with a branch: the padding is never executed, and edits of the entry code no longer move the loop), then the
// offset that measured best (-DPIN_NOPS=n: n dwords; the default puts .Lcmd where rounds 2 and 3 measured it best).
#ifndef PIN_NOPS
#ifdef BRX_WIN_SGPR
#define PIN_NOPS 5                                      // (the sparse-launch build: profiles/r03_ab.txt)
#else
#define PIN_NOPS 7
#endif
#endif
    .p2align 8
    .rept PIN_NOPS
    s_nop 0
    .endr
// .Lcmd: the lookup of the insert&copy symbol from scratch (the uncommon copy paths end here);
// .Lcmd_pre: entered from the common copy paths, which issued the lookup's compare and fetches at .Lcopy.
#ifdef BRX_NO_SPEC
.Lcmd_pre:
#endif
.Lcmd:
    PROF_MARK s31                                       // copy + tail
    s_sub_u32 IBLEN, IBLEN, 1
    s_cbranch_scc1 .Lx_r0_switch
.Lcmd_ticked:                                           // (back from an insert&copy block switch)
    LOOKUP2F VIACL, VIACB, 1, ds_read_u16, 3
#ifndef BRX_NO_SPEC
    s_branch .Lcmd_have
.Lcmd_pre:
    PROF_MARK s31                                       // copy + tail
    s_sub_u32 IBLEN, IBLEN, 1
    s_cbranch_scc1 .Lx_r0_switch
    LOOKUP2F_LATE 11
#endif
.Lcmd_have:
    s_waitcnt lgkmcnt(0)
    v_readlane_b32 T0, VS, CLEN                         // byte offset of the symbol's record in the insert&copy table
    s_load_dwordx4 s[92:95], IACTAB, T0                 // = INS base, CPY base, DCTX, extra-bit counts
    s_waitcnt lgkmcnt(0)
    s_cmp_lg_u32 s95, 0
    s_cbranch_scc1 .Liac_extras                         // (three commands in ten on text)

// ======================================================================================================== R1
.Lr1:
    PROF_MARK s23                                       // insert&copy symbol (+ extras; + entry)
    // the distance tree depends on the copy code only: request its limits / bases now, use them after the literals
#ifndef BRX_DIST_RESIDENT
    v_readlane_b32 DTREE, VDH4, DCTX
    s_nop 1
    v_add_u32 VT0, DTREE, VLANE4                        // (a one-symbol tree has no header: an out-of-range read, returns 0)
    ds_read_b32 VDHV, VT0
#endif
    s_cmp_lg_u32 INS, 0
    s_cbranch_scc1 .Lhave_lits                          // one command in three has literals
.Lno_lits:
    PROF_MARK s29                                       // R1 dispatch + literals
#ifdef BRX_DIST_RESIDENT
    // (the trees are resident: no descriptor needed on the common path.)  Five commands in six have a copy of 5 bytes or more,
    // distance context 3: its tree pair is looked up by name -- the two scalar instructions of the VGPR index mode are
    // the dearest ones here (DESIGN.md 4.1) -- the other contexts and the special trees go out of line (.Ldist_other)
    s_bitcmp1_b32 DSPEC, DCTX
    s_cbranch_scc1 .Ldist_other
#else
    s_cmp_lt_i32 DTREE, 0
    s_cbranch_scc1 .Ldist_special                       // implicit distance 0, or a one-symbol tree
#endif
    // ---- distance symbol (reference parse_distance_code :1367-1410)
    s_sub_u32 DBLEN, DBLEN, 1
    s_cbranch_scc1 .Lx_dist_switch
#ifdef BRX_DIST_RESIDENT
    LOOKUP2F VD3L, VD3B, 2, ds_read_b32, 5
.Ldist_have:
#else
.Ldist_ticked:                                          // (back from a distance block switch)
    s_waitcnt lgkmcnt(0)
    SPLIT_TREE VDHV, VDHB
    LOOKUP2 VDHV, VDHB, DTREE, 2, ds_read_b32, SYMOFF, 5
#endif
    s_waitcnt lgkmcnt(0)
    // payload of the distance symbol (decode_distance :1412-1481): extra-bit count | base << 5, or (bit 31) one of the 16
    // last-distance codes / a symbol without payload form (BRX_DIST_UNFIT: handed back with its bits un-taken)
    v_readlane_b32 DCODE, VS, CLEN
    s_cmp_lt_i32 DCODE, 0
    s_cbranch_scc1 .Ldist_ring
    s_and_b32 T1, DCODE, 31
#ifdef BRX_WIN_SGPR
    s_lshr_b32 T2, DCODE, 5
    TAKE_EXTRA DIST, T2, T1, 6, NPOST                      // base + (extra << NPOSTFIX)
#else
    v_lshrrev_b32 VT0, 5, DCODE                         // (the base stays on the vector side)
    TAKE_EXTRA DIST, VT0, T1, 6, NPOST                     // base + extra * (1 << NPOSTFIX)
#endif
.Ldist_push:
    PROF_MARK s30                                       // distance symbol
    s_cmp_gt_u32 DIST, MAXA                             // MAXA: min(POS, WINDOW) as of its last exact evaluation (a lower bound)
    s_cbranch_scc1 .Ldict_check                         // :1476 not pushed: static dictionary reference
.Ldist_push_ok:
    v_mov_b32_dpp VRING, VRING row_shr:1 row_mask:0xf bank_mask:0xf
    v_writelane_b32 VRING, DIST, 0

// ---- window copy of <= 64 bytes that does not overlap its source (copy_literals :1483-1542): takes the next CPY lanes
// of the pending register
.Lcopy:
    // ONE test for the common case: the copy's lanes -- PFREE + CPY of them in use afterwards -- must fit the pending register
    // (<= 63: s_bfm_b64 takes a 6-bit width) and must not exceed the distance: then the copy does not overlap itself
    // (CPY <= DIST) AND its source ends before the first pending byte (POS - DIST + CPY <= PBASE = POS - PFREE).  Everything
    // else -- lanes used up or a source that reaches into pending bytes (land, come back), overlap, long copies -- is out of
    // line.  The end of the meta-block is checked behind the copy (.Lcopy_tail).
#ifndef BRX_NO_SPEC
    LOOKUP2F_EARLY VIACL, VIACB, 1, ds_read_u16         // the NEXT command's insert&copy symbol (.Lcmd_pre)
#endif
    s_bfm_b64 exec, CPY, PFREE                          // the copy's lanes (only scalar instructions up to the address: the mask is
    s_add_u32 PFREE, PFREE, CPY                         // made while PFREE is still the old one, PFREE moves in place: .Lcopy_slow_undo)
    s_min_u32 T0, DIST, 63
    s_cmp_gt_u32 PFREE, T0
    s_cbranch_scc1 .Lcopy_slow_undo
    s_sub_u32 T2, PBASE, DIST                           // + lane = position of this lane's source byte  (PBASE = POS - PFREE)
    s_cmp_gt_u32 DIST, RING
    s_cbranch_scc0 .Lcopy_near
    // older than the ring: final in HBM (never pending bytes: what is pending is younger than one flush block)
    v_add_u32 VT0, T2, VLANE
    buffer_load_ubyte VPEND, VT0, RSRC, 0 offen
    s_mov_b64 exec, XLOOP
    s_add_u32 POS, POS, CPY
.Lcopy_tail_pre:
    s_cmp_ge_u32 POS, MBEND
    s_cbranch_scc0 .Lcmd_pre
    s_branch .Lcopy_end
.Lcopy_tail:                                            // (the flush cursor is checked whenever pending copies land)
.Lflush_back_cmd:
    s_cmp_ge_u32 POS, MBEND
    s_cbranch_scc0 .Lcmd
.Lcopy_end:
    s_cmp_eq_u32 POS, MBEND
    s_cbranch_scc0 .Lcopy_overrun
    s_mov_b32 INS, 0
    s_branch .Lexit
.Lcopy_overrun:                                         // :2105 the copy runs past the meta-block: take its lanes back (nothing
    s_sub_u32 POS, POS, CPY                             // of it has landed), the C++ side raises the error at R2
    s_sub_u32 PFREE, PFREE, CPY
    s_branch .Lx_r2

// ---- source inside the ring (final bytes: the test above keeps it clear of the pending ones)
.Lcopy_near:
    s_add_u32 T2, T2, SKEW
    v_add_u32 VT0, T2, VLANE
    v_and_b32 VT0, RMASK, VT0
    ds_read_u8 VPEND, VT0
    s_mov_b64 exec, XLOOP
    s_add_u32 POS, POS, CPY
    s_cmp_ge_u32 POS, MBEND
    s_cbranch_scc0 .Lcmd_pre
    s_branch .Lcopy_end

#ifdef BRX_DIST_RESIDENT
.Ldist_other:
    s_add_u32 T0, DCTX, 16
    s_bitcmp1_b32 DSPEC, T0
    s_cbranch_scc1 .Ldist_special
    s_sub_u32 DBLEN, DBLEN, 1
    s_cbranch_scc1 .Lx_dist_switch
.Ldist_ticked:                                          // (back from a distance block switch, too: any context)
    v_bfrev_b32 VR, WSRC
    v_lshrrev_b32 VI, VSH, VR
    s_set_gpr_idx_on DCTX, 6                            // SRC1 | SRC2 + 2 * context
    v_cmp_lt_u32 vcc, VI, VDTREES
    v_lshl_add_u32 VI, VI, 2, VDTREES1
    s_set_gpr_idx_off
    ds_read_b32 VS, VI
    s_ff1_i32_b32 CLEN, vcc_lo
    TAKE CLEN, 15
    s_branch .Ldist_have
#endif
.Ldist_special:
#ifdef BRX_DIST_RESIDENT
    v_readlane_b32 DTREE, VDH4, DCTX
#endif
    s_bitcmp1_b32 DTREE, 30
    s_cbranch_scc0 .Ldist_single
.Ldist_zero:
    v_readlane_b32 DIST, VRING, 0
    s_cmp_gt_u32 DIST, MAXA
    s_cbranch_scc0 .Lcopy
    s_min_u32 MAXA, POS, WINDOW
    s_cmp_gt_u32 DIST, MAXA
    s_cbranch_scc0 .Lcopy
    s_branch .Ldict
.Ldict_check:                                           // the exact bound
    s_min_u32 MAXA, POS, WINDOW
    s_cmp_gt_u32 DIST, MAXA
    s_cbranch_scc0 .Ldist_push_ok
    s_branch .Ldict

// ---- insert&copy extra bits (decode_insert_and_copy_length :1210-1224)
.Liac_extras:
    s_and_b32 T2, s95, 0xff                             // insert extra bits
    s_bfe_u32 T3, s95, 0x80008                          // copy extra bits
    TAKE_EXTRA INS, INS, T2, 1
    TAKE_EXTRA CPY, CPY, T3, 2
    s_branch .Lr1

// ---- literal runs (the register-resident loops)
// The common run is the whole insert: no block end, no flush block end inside it (a flush that falls due exactly at its
// end is taken by the next landing or run).  Anything else takes the general setup.
.macro LIT_RUN_FAST general
    s_sub_u32 T6, FLUSHAT, POS
    s_cbranch_scc1 \general
    s_min_u32 T6, T6, LBLEN
    s_cmp_gt_u32 INS, T6
    s_cbranch_scc1 \general
    s_sub_u32 LBLEN, LBLEN, INS
    s_add_u32 POS, POS, INS
    s_sub_u32 RUN, INS, 1
    s_mov_b32 INS, 0
.endm
.macro LIT_RUN_SETUP flush_stub, switch_stub
    s_cmp_eq_u32 LBLEN, 0
    s_cbranch_scc1 \switch_stub
    s_min_u32 RUN, INS, LBLEN
    s_sub_u32 T6, FLUSHAT, POS                          // (a copy may have ended exactly on the flush block: flush first)
    s_cbranch_scc1 \flush_stub
    s_cmp_eq_u32 T6, 0
    s_cbranch_scc1 \flush_stub
    s_min_u32 RUN, RUN, T6
    s_sub_u32 INS, INS, RUN
    s_sub_u32 LBLEN, LBLEN, RUN
    s_add_u32 POS, POS, RUN
    s_sub_u32 RUN, RUN, 1
    v_and_b32 VPA, RMASK, VPA                           // (the run before may have ended at the ring's end)
.endm
.macro LIT_RUN_END again, flush_stub
    s_cmp_ge_u32 POS, FLUSHAT
    s_cbranch_scc1 \flush_stub
    s_cmp_lg_u32 INS, 0
    s_cbranch_scc1 \again
    s_branch .Lafter_lits
.endm
// out of line: the flush at a run's end; the refill of a run's loop, which gives the untouched rest of the run back
// (INS, LBLEN, POS) before the input staging rolls -- that may poison the block counts or find the end of the input
.macro LIT_RUN_STUBS id, again, flush_stub
.Llsw_\id:                                              // the literal block is used up at a run's start: switch, set the run up again
    s_call_b64 LINKB, .Lsw_L
    s_branch \again
\flush_stub:
    s_call_b64 LINKC, .Lflush
    s_cmp_lg_u32 INS, 0
    s_cbranch_scc1 \again
    s_branch .Lafter_lits
.Lrf_stub_\id:
    REFILL_CORE
    s_cbranch_scc1 .Lrf_back_\id
    s_add_u32 INS, INS, RUN                             // the run ends with the literal in progress (its bits are taken):
    s_add_u32 LBLEN, LBLEN, RUN                         // RUN = the literals behind it
    s_sub_u32 POS, POS, RUN
    s_mov_b32 RUN, 0
    s_call_b64 LINKA, .Lspecial
    s_branch .Lrf_back_\id
.endm

.macro LIT_RUN_FAST_STUB id, general_id
.Lrf_stub_\id:
    REFILL_CORE
    s_cbranch_scc1 .Lrf_back_\id
    s_mov_b32 INS, RUN
    s_add_u32 LBLEN, LBLEN, RUN
    s_sub_u32 POS, POS, RUN
    s_mov_b32 RUN, 0
    s_call_b64 LINKA, .Lspecial
    s_branch .Lrf_back_\general_id
.endm

// (LAND_BODY: see "helpers" below, .Lland)
.macro LAND_BODY pbase=1
    s_add_u32 T6, PBASE, SKEW
    s_bfm_b64 exec, PFREE, 0
    v_add_u32 VT1, T6, VLANE
    v_and_b32 VT1, RMASK, VT1
    PROF_WAIT_VM
    s_waitcnt vmcnt(0) lgkmcnt(0)
    ds_write_b8 VT1, VPEND
    s_mov_b64 exec, XLOOP
    s_mov_b32 PFREE, 0
    .if \pbase
    s_mov_b32 PBASE, POS
    .endif
.endm
// ---- literals (reference parse_insert_literals :1286-1365)
.Lhave_lits:
    s_sub_u32 MBLEFT, MBEND, POS
    s_cmp_gt_u32 INS, MBLEFT
    s_cbranch_scc1 .Lexit                               // :2036, raised by the C++ side
#ifndef BRX_PROF
    s_setpc_b64 LITJ
.Llit_entry_g:
#endif
    s_bitcmp1_b32 FLAGS, 3
    s_cbranch_scc1 .Lhave_lits1
    s_call_b64 LINKB, .Lland_ctx                        // pending bytes into the ring, context of the first literal, VPA
    s_bitcmp1_b32 FLAGS, 5
    s_cbranch_scc1 .Llit_r_start
    s_sub_u32 INS, INS, 1                               // the loop counts down to the borrow
    s_bitcmp1_b32 FLAGS, 4
    s_cbranch_scc1 .Llit_m
// \mixed = 0: the entries carry the context info; 1 (meta-blocks whose literal block types differ in context mode): they
// are plain bytes and the info comes from the table of the current block type's mode
.macro LIT_LOOP sfx, mixed, rid
.Llit\sfx:
    s_sub_u32 LBLEN, LBLEN, 1
    s_cbranch_scc1 .Lx_lit_switch\sfx
.Llit_ticked\sfx:
    s_waitcnt lgkmcnt(0)                                // VH = tree descriptor of this literal's context
    v_cmp_gt_i32 vcc, 0, VH
    s_cbranch_vccnz .Llit_single\sfx
    v_add_u32 VT0, VH, VLANE4
    ds_read_b32 VLIM, VT0
    s_waitcnt lgkmcnt(0)
    SPLIT_TREE VLIM, VBASE
    LOOKUP2 VLIM, VBASE, VH, 1, ds_read_u16, SYMOFF, \rid
    s_waitcnt lgkmcnt(0)
    v_readlane_b32 T0, VS, CLEN                         // byte | context info << 8
    v_and_b32 VT0, RMASK, VPA
    s_nop 0
    v_mov_b32 VE, T0
    ds_write_b8 VT0, VE
.Llit_stored\sfx:
.if \mixed
    v_and_b32 VT1, 0xff, VE
    ds_read_u8 VINFO, VT1 offset:LDS_ITAB
    s_waitcnt lgkmcnt(0)
.else
    v_lshrrev_b32 VINFO, 8, VE
.endif
    v_and_or_b32 VC, VINFO, MA, VB4                     // context id * 4 of the next literal
    ds_read_b32 VH, VC offset:LDS_CMH
    v_and_b32 VB4, MB, VINFO
    v_lshlrev_b32 VB4, SB, VB4                          // this literal's share as p2 of the one after
    v_add_u32 VPA, 1, VPA
    s_add_u32 POS, POS, 1
    s_cmp_ge_u32 POS, FLUSHAT
    s_cbranch_scc1 .Lflush_stub_lit\sfx
.Lflush_back_lit\sfx:
    s_sub_u32 INS, INS, 1
    s_cbranch_scc0 .Llit\sfx
    s_branch .Lafter_lits
.Llit_single\sfx:                                       // one-symbol tree: no bits
    v_and_b32 VE, 0xffff, VH
    v_and_b32 VT0, RMASK, VPA
    ds_write_b8 VT0, VE
    s_branch .Llit_stored\sfx
.Lflush_stub_lit\sfx:
    s_call_b64 LINKC, .Lflush
    s_branch .Lflush_back_lit\sfx
.endm
    LIT_LOOP _u, 0, 4
    LIT_LOOP _m, 1, 8
// resident trees: context arithmetic on the scalar side (the entry arrives in an SGPR anyway; the same on the vector side --
// three uniform VALU instructions + v_readfirstlane for four SALU -- measured slower at every loop position, profiles/r03_ab.txt),
// tree pair through the VGPR index mode
#ifdef BRX_PROF
.Llit_r_start:
    v_lshrrev_b32 VT0, 2, VC                            // (this loop works on the context id itself, not id * 4)
    v_lshrrev_b32 VT1, 2, VB4
    s_nop 0
    v_readfirstlane_b32 T4, VT0                         // context id of the first literal
    v_readfirstlane_b32 T5, VT1                         // p1's share as a later p2
.Llit_r_go:
// A run = literals up to the end of the insert, of the literal block, or of the flush block, whichever is first: INS,
// LBLEN and POS move once per run, the loop itself counts RUN down (one behind: to the borrow).
.macro LIT_R_BODY rid
    v_readlane_b32 T6, VCMAP, T4                        // 2 * tree index
    LOOKUP2X \rid
    s_waitcnt lgkmcnt(0)
    v_readlane_b32 T0, VS, CLEN                         // byte | context info << 8
    s_bfe_u32 T1, T0, 0x6000a                           // context info of this literal >> 2
    s_and_b32 T1, T1, MA2
    v_mov_b32 VE, T0                                    // (two instructions behind the v_readlane: VALU-written SGPR)
    ds_write_b8 VPA, VE                                 // (VPA is a ring address and a run ends at the flush block's end at the latest)
    v_add_u32 VPA, 1, VPA
    s_or_b32 T4, T1, T5                                 // context id of the next one
    s_bfe_u32 T5, T0, BFEB                              // ... and this literal's share of the one after, as a field of the entry
.endm
    LIT_RUN_FAST .Llit_r_run
.Llit_rf:
    LIT_R_BODY 10
    s_sub_u32 RUN, RUN, 1
    s_cbranch_scc0 .Llit_rf
.Lafter_lits_fast:
    s_mov_b32 PBASE, POS
    s_cmp_eq_u32 POS, MBEND
    s_cbranch_scc0 .Lno_lits
    s_branch .Lexit
.Llit_r_run:
    LIT_RUN_SETUP .Lflush_stub_lit_r, .Llsw_9
.Llit_r:
    LIT_R_BODY 9
    s_sub_u32 RUN, RUN, 1
    s_cbranch_scc0 .Llit_r
    LIT_RUN_END .Llit_r_run, .Lflush_stub_lit_r
    LIT_RUN_STUBS 9, .Llit_r_run, .Lflush_stub_lit_r
    LIT_RUN_FAST_STUB 10, 9
#else
// Two variants of these loops, picked at entry through LITJ:
//   m3  context mode 3 (signed): id = lut2(p1) << 3 | lut2(p2), and a literal's share as p1 and as p2 is the same 3-bit value --
//       one s_bfe + one s_lshl3_add_u32 per literal, the loop unrolled twice so that "this" and "previous" swap registers
//       (T7 / T5: both survive a refill and .Lspecial; T1 does not in the sparse-launch build) instead of being copied
//   mx  the other modes: the six bits of the p1 share need no mask there (MA2 = 0x3f)
// (.Lhave_lits jumps to .Llit_r_entry_*) pending bytes into the ring, then the context of the first literal straight into the
// scalar registers of the loop -- .Lland_ctx without the detour through the vector-side form
// The two bytes in front of a run that follows a copy are the LAST TWO PENDING ones (a copy is at least 2 bytes long, a
// dictionary word 4): they are read from their lanes of VPEND, and their context info from its lane of VITAB (4 bytes per
// lane) -- no trip to the ring and none to the info table in front of the run's first literal (two dependent LDS round
// trips less per run; the landing's store is no longer waited for by anything).  -DBRX_CTX_RING: the bytes from the ring.
.macro LIT_CTX_RING_BYTES                               // T6 / T7 = context info of the last two bytes of the ring
    v_add_u32 VT0, -1, VPA
    v_add_u32 VT1, -2, VPA
    v_and_b32 VT0, RMASK, VT0
    v_and_b32 VT1, RMASK, VT1
    ds_read_u8 VT0, VT0
    ds_read_u8 VT1, VT1
    s_waitcnt lgkmcnt(0)
    ds_read_u8 VT3, VT0 offset:LDS_ITAB                 // info(p1)
    ds_read_u8 VT4, VT1 offset:LDS_ITAB                 // info(p2)
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 T6, VT3
    v_readfirstlane_b32 T7, VT4
.endm
.macro LIT_CTX_ENTRY v
.Llit_r_entry_\v:
    s_cmp_eq_u32 PFREE, 0
    s_cbranch_scc1 .Llit_r_entry_nopend_\v               // (out of line, LIT_CTX_ENTRY_AUX: only there can this be the stream's start)
#ifdef BRX_CTX_RING
    LAND_BODY 0                                         // (PBASE: the run's end sets it, nothing reads it in between)
.Llit_r_entry_ring_\v:
    s_add_u32 T6, POS, SKEW
    v_bfe_u32 VPA, T6, 0, 11
    LIT_CTX_RING_BYTES
#else
    s_sub_u32 T4, PFREE, 1                              // lanes of the last two pending bytes
    s_sub_u32 T5, PFREE, 2
    LAND_BODY 0                                         // (PBASE: the run's end sets it, nothing reads it in between)
    v_readlane_b32 T6, VPEND, T4                        // p1 (the landing has waited for the copies' loads)
    v_readlane_b32 T7, VPEND, T5                        // p2
    s_add_u32 T0, POS, SKEW
    v_bfe_u32 VPA, T0, 0, 11
    s_lshr_b32 T0, T6, 2                                // info(p) = byte p & 3 of lane p >> 2 of VITAB
    s_lshr_b32 T1, T7, 2
    v_readlane_b32 T0, VITAB, T0
    v_readlane_b32 T1, VITAB, T1
    s_lshl_b32 T6, T6, 3                                // (a shift takes the low five bits of its count: 8 * (p & 3))
    s_lshl_b32 T7, T7, 3
    s_lshr_b32 T6, T0, T6                               // (bits 8.. are the neighbours' info: only fields of bits 7:0 are taken below)
    s_lshr_b32 T7, T1, T7
#endif
.Llit_r_ctx_have_\v:
    s_bfe_u32 T1, T6, 0x60002
    s_and_b32 T1, T1, MA2
    s_bfe_u32 T5, T6, BFEBI                             // p1's share as a later p2
    s_bfe_u32 T7, T7, BFEBI
    s_or_b32 T4, T1, T7                                 // context id of the first literal
.Llit_r_go_\v:
.endm
.macro LIT_CTX_ENTRY_AUX v
#ifndef BRX_CTX_RING
.Llit_r_entry_ring_\v:                                  // nothing pending (a landing, a flush or a long copy came in between): the ring has the bytes
    s_add_u32 T6, POS, SKEW
    v_bfe_u32 VPA, T6, 0, 11
    LIT_CTX_RING_BYTES
    s_branch .Llit_r_ctx_have_\v
#endif
.Llit_r_entry_nopend_\v:
    s_cmp_lt_u32 POS, 2                                 // (a copy is at least 2 bytes long: nothing is pending at the stream's start)
    s_cbranch_scc0 .Llit_r_entry_ring_\v
    s_call_b64 LINKB, .Lland_ctx                        // the first two bytes of a stream
.ifc \v,mx
.Llit_r_start:
.endif
    v_lshrrev_b32 VT0, 2, VC                            // (this loop works on the context id itself, not id * 4)
    v_lshrrev_b32 VT1, 2, VB4
    s_nop 0
    v_readfirstlane_b32 T4, VT0                         // context id of the first literal
    v_readfirstlane_b32 T5, VT1                         // p1's share as a later p2
    s_branch .Llit_r_go_\v
.endm
// A run = literals up to the end of the insert, of the literal block, or of the flush block, whichever is first: INS,
// LBLEN and POS move once per run, the loop itself counts RUN down (one behind: to the borrow).
// One literal: T4 = its context id.  \cur = register that takes this literal's share, \prev = the one holding the previous one's.
.macro LIT_R_HEAD rid
    v_readlane_b32 T6, VCMAP, T4                        // 2 * tree index
    LOOKUP2X \rid
    s_waitcnt lgkmcnt(0)
    v_readlane_b32 T0, VS, CLEN                         // byte | context info << 8
.endm
.macro LIT_R_STORE
    v_mov_b32 VE, T0                                    // (two instructions behind the v_readlane: VALU-written SGPR)
    ds_write_b8 VPA, VE                                 // (VPA is a ring address and a run ends at the flush block's end at the latest)
    v_add_u32 VPA, 1, VPA
.endm
.macro LIT_R_BODY_M3 rid, cur, prev
    LIT_R_HEAD \rid
    s_bfe_u32 \cur, T0, 0x3000d                         // lut2 of this literal (context info bits 7:5 = bits 4:2)
    s_lshl3_add_u32 T4, \cur, \prev                     // context id of the next one
    LIT_R_STORE
.endm
.macro LIT_R_BODY_MX rid
    LIT_R_HEAD \rid
    s_bfe_u32 T1, T0, 0x6000a                           // context info of this literal >> 2: its share as p1
    s_or_b32 T4, T1, T5                                 // context id of the next one
    LIT_R_STORE
    s_bfe_u32 T5, T0, BFEB                              // ... and this literal's share of the one after, as a field of the entry
.endm
.macro LIT_RF_STUB id, back_id                          // (the refill part of LIT_RUN_STUBS / LIT_RUN_FAST_STUB)
.Lrf_stub_\id:
    REFILL_CORE
    s_cbranch_scc1 .Lrf_back_\id
    s_add_u32 INS, INS, RUN                             // the run ends with the literal in progress (its bits are taken):
    s_add_u32 LBLEN, LBLEN, RUN                         // RUN = the literals behind it
    s_sub_u32 POS, POS, RUN
    s_mov_b32 RUN, 0
    s_call_b64 LINKA, .Lspecial
    s_branch .Lrf_back_\back_id
.endm
.macro LIT_RUN_TAIL_FAST
    s_mov_b32 PBASE, POS
    s_cmp_eq_u32 POS, MBEND
    s_cbranch_scc0 .Lno_lits
    s_branch .Lexit
.endm
.macro LIT_RUN_AUX v                                    // (the out-of-line parts of LIT_RUN_STUBS that are not refills)
.Llsw_\v:                                               // the literal block is used up at a run's start: switch, set the run up again
    s_call_b64 LINKB, .Lsw_L
    s_branch .Llit_r_run_\v
.Lflush_stub_lit_r_\v:
    s_call_b64 LINKC, .Lflush
    s_cmp_lg_u32 INS, 0
    s_cbranch_scc1 .Llit_r_run_\v
    s_branch .Lafter_lits
.endm
// ---- m3
    LIT_CTX_ENTRY m3
    LIT_RUN_FAST .Llit_r_run_m3
.Llit_rf_m3:
    LIT_R_BODY_M3 40, T7, T5
    s_sub_u32 RUN, RUN, 1
    s_cbranch_scc1 .Lafter_lits_fast_m3
    LIT_R_BODY_M3 41, T5, T7
    s_sub_u32 RUN, RUN, 1
    s_cbranch_scc0 .Llit_rf_m3
.Lafter_lits_fast_m3:
    LIT_RUN_TAIL_FAST
.Llit_r_run_m3:
    LIT_RUN_SETUP .Lflush_stub_lit_r_m3, .Llsw_m3
.Llit_r_m3:
    LIT_R_BODY_M3 42, T7, T5
    s_sub_u32 RUN, RUN, 1
    s_cbranch_scc1 .Llit_r_m3_odd
    LIT_R_BODY_M3 43, T5, T7
    s_sub_u32 RUN, RUN, 1
    s_cbranch_scc0 .Llit_r_m3
.Llit_r_m3_end:
    LIT_RUN_END .Llit_r_run_m3, .Lflush_stub_lit_r_m3
.Llit_r_m3_odd:                                         // (the next run starts with the previous share in T5 again)
    s_mov_b32 T5, T7
    s_branch .Llit_r_m3_end
    LIT_RUN_AUX m3
    LIT_CTX_ENTRY_AUX m3
    LIT_RF_STUB 40, 42
    LIT_RF_STUB 41, 43
    LIT_RF_STUB 42, 42
    LIT_RF_STUB 43, 43
// ---- mx
    LIT_CTX_ENTRY mx
    LIT_RUN_FAST .Llit_r_run_mx
.Llit_rf_mx:
    LIT_R_BODY_MX 44
    s_sub_u32 RUN, RUN, 1
    s_cbranch_scc0 .Llit_rf_mx
    LIT_RUN_TAIL_FAST
.Llit_r_run_mx:
    LIT_RUN_SETUP .Lflush_stub_lit_r_mx, .Llsw_mx
.Llit_r_mx:
    LIT_R_BODY_MX 45
    s_sub_u32 RUN, RUN, 1
    s_cbranch_scc0 .Llit_r_mx
    LIT_RUN_END .Llit_r_run_mx, .Lflush_stub_lit_r_mx
    LIT_RUN_AUX mx
    LIT_CTX_ENTRY_AUX mx
    LIT_RF_STUB 44, 45
    LIT_RF_STUB 45, 45
#ifdef BRX_SLOTS
// ---- the pipelined loop over the tree cache.  One compare + one fetch (SL_COMPARE) decode the literal at the window's low end
// under all four cached trees at once: group s of 16 lanes = slot s, its compare bits = bits 16 s .. of the mask, its
// candidates' entries fetched per lane as in LOOKUP2F.  SL_SELECT picks the group of the literal's context (VSLOT), takes
// the code's bits, and the NEXT literal's SL_COMPARE is issued before this literal's entry is waited for (SL_FINISH: entry,
// context of the next literal).  Two literals are in flight (VSA / VSB, VCCA / VCCB: the loop is unrolled twice).
// The entries are collected in VLITS and go to the ring with one store at the run's end.  A context whose tree is not
// cached leaves through \miss: the tree replaces the oldest one, the compare is issued again.
// A run is at most 63 literals and at most what the input is sure to hold in front of its last END_MARGIN dwords (2 literals per
// dword: a literal's code is <= 15 bits) -- inside a run .Lspecial only ever rolls the staging, EXEC = all lanes throughout;
// the literals next to the end of the input take the per-literal loop.
.macro SL_COMPARE vs, vccp
    v_bfrev_b32 VR, WINLO
    v_lshrrev_b32 VI, VSH, VR
    v_cmp_lt_u32 \vccp, VI, VQL
    v_lshl_add_u32 VI, VI, 1, VQB
    ds_read_u16 \vs, VI
.endm
.macro SL_SELECT vccp, rid, miss
    v_readlane_b32 T6, VSLOT, T4                        // 16 * slot of this literal's tree
    s_bitcmp1_b32 T6, 7
    s_cbranch_scc1 \miss
    s_lshr_b64 T01, \vccp, T6
    s_ff1_i32_b32 CLEN, T0                              // code length = the group's lowest matching lane
    s_add_u32 T6, T6, CLEN                              // lane of the entry
    TAKE CLEN, \rid
.endm
.macro SL_FINISH_M3 vs, cnt, cur, prev
    s_waitcnt lgkmcnt(\cnt)
    v_readlane_b32 T0, \vs, T6                          // byte | context info << 8
    s_bfe_u32 \cur, T0, 0x3000d                         // lut2 of this literal (context info bits 7:5 = bits 4:2)
    s_lshl3_add_u32 T4, \cur, \prev                     // context id of the next one
    v_writelane_b32 VLITS, T0, m0
.endm
.macro SL_FINISH_MX vs, cnt
    s_waitcnt lgkmcnt(\cnt)
    v_readlane_b32 T0, \vs, T6
    s_bfe_u32 T1, T0, 0x6000a                           // context info of this literal >> 2: its share as p1
    s_or_b32 T4, T1, T5                                 // context id of the next one
    s_bfe_u32 T5, T0, BFEB                              // ... and this literal's share of the one after, as a field of the entry
    v_writelane_b32 VLITS, T0, m0
.endm
.macro SL_RF_STUB id                                    // (lane WLSTOP inside a run is the end of the staged chunk, never the end
.Lrf_stub_\id:                                          // of the input: SL_RUN; .Lspecial rolls the staging and nothing else)
    REFILL_CORE
    s_cbranch_scc1 .Lrf_back_\id
    s_call_b64 LINKA, .Lspecial
    s_mov_b64 exec, -1
    s_branch .Lrf_back_\id
.endm
.macro SL_MISS v, ph, vs, vccp
.Lsl_miss_\ph\()_\v:
    v_readlane_b32 T6, VCMAP, T4
    s_lshr_b32 T6, T6, 1
    s_call_b64 LINKB, .Lslot_fill
    SL_COMPARE \vs, \vccp
    s_branch .Lsl_\ph\()_\v
.endm
// the run's setup (RUN = number of literals - 1; M0 counts them down), its end, its out-of-line decisions
.macro SL_RUN v
.Llit_r_run_\v:
    s_sub_u32 T6, FLUSHAT, POS                          // (a copy may have ended exactly on the flush block: flush first)
    s_cbranch_scc1 .Lflush_stub_lit_r_\v
    s_min_u32 T6, T6, LBLEN
    s_add_u32 T0, CBASE, WL
    s_add_u32 T0, T0, 1
    s_sub_u32 T0, WSAFE, T0                             // refills before the one that pulls in dword WSAFE - 1 (that one poisons the loop) ...
    s_cselect_b32 T0, 0, T0
    s_lshl_b32 T0, T0, 1                                // ... each good for two literals (a literal's code is <= 15 bits)
    s_min_u32 T0, T0, 63
    s_min_u32 T6, T6, T0
    s_min_u32 RUN, INS, T6
    s_cmp_eq_u32 T6, 0
    s_cbranch_scc1 .Lsl_slow_\v
    s_sub_u32 INS, INS, RUN
    s_sub_u32 LBLEN, LBLEN, RUN
    s_add_u32 POS, POS, RUN
    s_sub_u32 RUN, RUN, 1
    s_mov_b32 m0, RUN                                   // the loop counts M0 down (the lane select of v_writelane next to an SGPR operand)
    s_mov_b64 exec, -1
    SL_COMPARE VSA, VCCA
.endm
.macro SL_RUN_END v
.Lsl_end_\v:
    // literal k of the run sits in lane (RUN - 1 - k) & 63 of VLITS, k = 0 .. RUN: lanes 0 .. RUN - 1 and lane 63
    s_bfm_b64 exec, RUN, 0
    s_bitset1_b32 exec_hi, 31
    s_add_u32 T0, POS, SKEW
    s_sub_u32 T0, T0, RUN
    s_sub_u32 T0, T0, 1                                 // ring position of the run's first literal
    v_add_u32 VT0, RUN, VNLANE
    v_and_b32 VT0, 63, VT0
    v_add_u32 VT0, T0, VT0
    v_and_b32 VT0, RMASK, VT0
    ds_write_b8 VT0, VLITS
    s_mov_b64 exec, XLOOP
    LIT_RUN_END .Llit_r_run_\v, .Lflush_stub_lit_r_\v
.Lsl_slow_\v:
    s_cmp_eq_u32 LBLEN, 0
    s_cbranch_scc1 .Llsw_\v
    s_cmp_ge_u32 POS, FLUSHAT
    s_cbranch_scc1 .Lflush_stub_lit_r_\v
    // the end of the input is near: the rest of this insert takes the per-literal loop (it knows how to leave when the block
    // counters get poisoned) -- its context state is the vector-side form of T4 / T5
    s_lshl_b32 T0, T4, 2
    s_lshl_b32 T1, T5, 2
    v_mov_b32 VC, T0
    v_mov_b32 VB4, T1
    ds_read_b32 VH, VC offset:LDS_CMH
    s_add_u32 T0, POS, SKEW
    v_bfe_u32 VPA, T0, 0, 11
    s_sub_u32 INS, INS, 1
    s_branch .Llit_u
.endm
// ---- context mode 3 (T7 / T5: this literal's share and the previous one's swap roles from literal to literal)
    LIT_CTX_ENTRY s3
    SL_RUN s3
.Lsl_a_s3:
    SL_SELECT VCCA, 50, .Lsl_miss_a_s3
    s_sub_u32 m0, m0, 1
    s_cbranch_scc1 .Lsl_last_a_s3
    SL_COMPARE VSB, VCCB
    SL_FINISH_M3 VSA, 1, T7, T5
.Lsl_b_s3:
    SL_SELECT VCCB, 51, .Lsl_miss_b_s3
    s_sub_u32 m0, m0, 1
    s_cbranch_scc1 .Lsl_last_b_s3
    SL_COMPARE VSA, VCCA
    SL_FINISH_M3 VSB, 1, T5, T7
    s_branch .Lsl_a_s3
.Lsl_last_a_s3:
    SL_FINISH_M3 VSA, 0, T7, T5
    s_mov_b32 T5, T7                                    // (the next run starts with the previous share in T5 again)
    s_branch .Lsl_end_s3
.Lsl_last_b_s3:
    SL_FINISH_M3 VSB, 0, T5, T7
    SL_RUN_END s3
    SL_MISS s3, a, VSA, VCCA
    SL_MISS s3, b, VSB, VCCB
    LIT_RUN_AUX s3
    LIT_CTX_ENTRY_AUX s3
    SL_RF_STUB 50
    SL_RF_STUB 51
// ---- the other modes
    LIT_CTX_ENTRY sx
    SL_RUN sx
.Lsl_a_sx:
    SL_SELECT VCCA, 52, .Lsl_miss_a_sx
    s_sub_u32 m0, m0, 1
    s_cbranch_scc1 .Lsl_last_a_sx
    SL_COMPARE VSB, VCCB
    SL_FINISH_MX VSA, 1
.Lsl_b_sx:
    SL_SELECT VCCB, 53, .Lsl_miss_b_sx
    s_sub_u32 m0, m0, 1
    s_cbranch_scc1 .Lsl_last_b_sx
    SL_COMPARE VSA, VCCA
    SL_FINISH_MX VSB, 1
    s_branch .Lsl_a_sx
.Lsl_last_a_sx:
    SL_FINISH_MX VSA, 0
    s_branch .Lsl_end_sx
.Lsl_last_b_sx:
    SL_FINISH_MX VSB, 0
    SL_RUN_END sx
    SL_MISS sx, a, VSA, VCCA
    SL_MISS sx, b, VSB, VCCB
    LIT_RUN_AUX sx
    LIT_CTX_ENTRY_AUX sx
    SL_RF_STUB 52
    SL_RF_STUB 53
#endif
#endif
.Lafter_lits:
    s_mov_b32 INS, 0
    s_mov_b32 PBASE, POS                                // (nothing is pending here)
    s_cmp_eq_u32 POS, MBEND
    s_cbranch_scc0 .Lno_lits
    s_branch .Lexit                                     // :2069 the copy part of the last command is ignored

// one literal tree, resident: no contexts
.Lhave_lits1:
    s_call_b64 LINKB, .Lland
    s_add_u32 T0, POS, SKEW
    v_bfe_u32 VPA, T0, 0, 11
    LIT_RUN_FAST .Llit1_run
    s_branch .Llit1
.Llit1_run:
    LIT_RUN_SETUP .Lflush_stub_lit1, .Llsw_7
#if defined(BRX_PROF) || defined(BRX_NO_SPEC) || defined(BRX_LIT1_SERIAL)
.Llit1:
    LOOKUP2F VLITL, VLITB, 1, ds_read_u16, 7
    s_waitcnt lgkmcnt(0)
    v_readlane_b32 T0, VS, CLEN
    s_nop 1
    v_mov_b32 VE, T0
    ds_write_b8 VPA, VE
    v_add_u32 VPA, 1, VPA
    s_sub_u32 RUN, RUN, 1
    s_cbranch_scc0 .Llit1
    s_cmp_lg_u32 INS, 0
    s_cbranch_scc0 .Lafter_lits
    LIT_RUN_END .Llit1_run, .Lflush_stub_lit1
    LIT_RUN_STUBS 7, .Llit1_run, .Lflush_stub_lit1
#else
// Round 5: pipelined.  Without a context nothing of literal k + 1 depends on literal k's ENTRY, only on its length: the next
// compare + fetch go out right behind the TAKE, and the entry is stored a literal later straight from its lane (EXEC = that one
// lane: no v_readlane / v_mov) -- the LDS round trip and a VALU -> SALU hand-over leave the chain (tools/ubench/litloop.hip:
// k_one_base / k_one_pipe_x).  VSA / VSB, T6 / T7: entries and lengths of the two literals in flight.  The refill stubs give the
// rest of a run back as before (RUN = the literals behind the one in progress holds at both TAKEs).
.macro LOOKUP1 vs, len, rid
    v_bfrev_b32 VR, WSRC
    v_lshrrev_b32 VI, VSH, VR
    v_cmp_lt_u32 vcc, VI, VLITL
    v_lshl_add_u32 VI, VI, 1, VLITB
    ds_read_u16 \vs, VI
    s_ff1_i32_b32 \len, vcc_lo
    TAKE \len, \rid
.endm
.macro STORE1 vs, len, cnt, off, step                   // (a run never wraps the ring: the second literal of a pair goes to offset 1)
    s_waitcnt lgkmcnt(\cnt)
    s_lshl_b32 exec_lo, 1, \len
    ds_write_b8 VPA, \vs offset:\off
    s_mov_b32 exec_lo, XLOOP
    .if \step
    v_add_u32 VPA, \step, VPA
    .endif
.endm
.Llit1:
    LOOKUP1 VS, T6, 7
.Llit1_a:
    s_sub_u32 RUN, RUN, 1
    s_cbranch_scc1 .Llit1_last_a
    LOOKUP1 VE, T7, 16
    STORE1 VS, T6, 1, 0, 0
    s_sub_u32 RUN, RUN, 1
    s_cbranch_scc1 .Llit1_last_b
    LOOKUP1 VS, T6, 17
    STORE1 VE, T7, 1, 1, 2
    s_branch .Llit1_a
.Llit1_last_b:
    STORE1 VE, T7, 0, 1, 2
    s_branch .Llit1_end
.Llit1_last_a:
    STORE1 VS, T6, 0, 0, 1
.Llit1_end:
    s_cmp_lg_u32 INS, 0
    s_cbranch_scc0 .Lafter_lits
    LIT_RUN_END .Llit1_run, .Lflush_stub_lit1
    LIT_RUN_STUBS 7, .Llit1_run, .Lflush_stub_lit1
    LIT_RF_STUB 16, 16
    LIT_RF_STUB 17, 17
#endif

// ---- last-distance codes 0..15 (decode_distance :1412-1450)
.Ldist_single:
    s_sub_u32 DBLEN, DBLEN, 1
    s_cbranch_scc1 .Lx_dist_switch
    s_and_b32 DCODE, DTREE, 0xffff
    s_branch .Ldist_ring_s
.Ldist_ring:
    s_bitcmp1_b32 DCODE, 30
    s_cbranch_scc1 .Lx_dist_unfit
    s_and_b32 DCODE, DCODE, 0xffff
.Ldist_ring_s:
    s_cmp_eq_u32 DCODE, 0
    s_cbranch_scc1 .Ldist_zero
    v_readlane_b32 T5, VRING, 1
    v_readlane_b32 T6, VRING, 2
    v_readlane_b32 T7, VRING, 3
    s_cmp_ge_u32 DCODE, 4
    s_cbranch_scc1 .Ldist_delta
    s_mov_b32 DIST, T5
    s_cmp_eq_u32 DCODE, 2
    s_cselect_b32 DIST, T6, DIST
    s_cmp_eq_u32 DCODE, 3
    s_cselect_b32 DIST, T7, DIST
    s_branch .Ldist_push
.Ldist_delta:
    v_readlane_b32 T0, VRING, 0
    s_cmp_lt_u32 DCODE, 10
    s_cselect_b32 T0, T0, T5
    s_cselect_b32 T1, 2, 8
    s_sub_u32 T1, DCODE, T1
    s_lshr_b32 T1, T1, 1
    s_sub_u32 T2, 0, T1
    s_bitcmp1_b32 DCODE, 0
    s_cselect_b32 T1, T1, T2
    s_add_i32 DIST, T0, T1
    s_cmp_le_i32 DIST, 0
    s_cbranch_scc1 .Lx_dist_bad
    s_branch .Ldist_push

// ---- static dictionary word (src/lib.rs:1506-1540); identity transform here, the others in .Ldict_xform
.Ldict:
    s_cmp_lt_u32 CPY, 4
    s_cbranch_scc1 .Lx_r2
    s_cmp_gt_u32 CPY, 24
    s_cbranch_scc1 .Lx_r2
    v_readlane_b32 T0, VDICTINFO, CPY                   // DOFFSET | NDBITS << 24
    s_sub_u32 T1, DIST, MAXA
    s_sub_u32 T1, T1, 1                                 // word id
    s_lshr_b32 T2, T0, 24
    s_lshr_b32 T3, T1, T2                               // transform id
    s_bfm_b32 T4, T2, 0
    s_and_b32 T1, T1, T4                                // word index
    s_and_b32 T0, T0, 0xffffff
    s_mul_i32 T1, T1, CPY
    s_add_u32 T0, T0, T1                                // byte offset of the word in the dictionary
    s_sub_u32 MBLEFT, MBEND, POS
    s_cmp_lg_u32 T3, 0
    s_cbranch_scc1 .Ldict_xform
    s_cmp_gt_u32 CPY, MBLEFT
    s_cbranch_scc1 .Lx_r2
    s_add_u32 T1, PFREE, CPY
    s_cmp_gt_u32 T1, 63
    s_cbranch_scc0 .Ldict_go
    s_call_b64 LINKB, .Lland
.Ldict_go:
    s_sub_u32 T2, T0, PFREE
    s_bfm_b64 exec, CPY, PFREE
    v_add_u32 VT0, T2, VLANE
    global_load_ubyte VPEND, VT0, DICTP
    s_mov_b64 exec, XLOOP
    s_add_u32 PFREE, PFREE, CPY
    s_add_u32 POS, POS, CPY
    s_branch .Lcopy_tail                                // (no early lookup on this path: it does not come through .Lcopy)

// ======================================================================================================== helpers
// Land the pending copies: lanes 0..PFREE-1 of VPEND hold the bytes of stream positions PBASE.. (PBASE = POS - PFREE;
// literals never sit between pending copies: a literal run lands everything first).  One masked byte store.
// Clobbers T6, VT1.  .Lland_ctx also derives the literal context from the last two bytes of the stream (VC = id * 4,
// VB4 = p1's share as a future p2) and requests the tree descriptor of the first literal.
.macro FLUSH_BODY lbl
    s_mov_b64 exec, -1
\lbl:
    s_and_b32 T6, VFL, RMASK
    v_add_u32 VT4, T6, VLANE16
    ds_read_b128 VQ, VT4
    s_sub_u32 T7, VFL, SKEW
    v_add_u32 VT4, T7, VLANE16
    s_waitcnt lgkmcnt(0)
    buffer_store_dwordx4 VQ, VT4, RSRC, 0 offen
#ifndef BRX_PROF
    buffer_store_dwordx4 VQ, VT4, RSRC2, 0 offen        // (the device-to-host copy rides on the decode)
#endif
    s_add_u32 VFL, VFL, 1024
    s_add_u32 FLUSHAT, FLUSHAT, 1024
    s_cmp_ge_u32 POS, FLUSHAT
    s_cbranch_scc1 \lbl
    s_mov_b64 exec, XLOOP
.endm
#ifdef BRX_SLOTS
// Literal tree T6 takes the next slot of the cache (round robin): the contexts of the tree that leaves are marked 0x80 in VSLOT,
// those of the new one get its slot; its header words -> limits as counts / folded bases in the slot's 16 lanes (as the entry
// code does for a resident tree).  Clobbers T0, T1, T6, CLEN, VT0, VT1, vcc; returns with EXEC = all lanes.
.Lslot_fill:
    s_mov_b64 exec, -1
    s_bfe_u32 T1, FLAGS, 0x20010                        // the slot
    s_add_u32 FLAGS, FLAGS, 0x10000
    s_bitset0_b32 FLAGS, 18
    s_lshl_b32 CLEN, T1, 3
    s_lshr_b32 T0, SLOTT, CLEN
    s_and_b32 T0, T0, 0xff                              // the tree that leaves (0xff: none)
    s_lshl_b32 T0, T0, 1                                // (VCMAP holds 2 * tree index; 0x1fe matches nothing)
    v_mov_b32 VT0, 0x80
    v_cmp_eq_u32 vcc, T0, VCMAP
    v_cndmask_b32 VSLOT, VSLOT, VT0, vcc
    s_lshl_b32 T0, T6, 1
    s_lshl_b32 T1, T1, 4
    v_mov_b32 VT0, T1
    v_cmp_eq_u32 vcc, T0, VCMAP
    v_cndmask_b32 VSLOT, VSLOT, VT0, vcc
    s_lshl_b32 T0, 0xff, CLEN
    s_andn2_b32 SLOTT, SLOTT, T0
    s_lshl_b32 T0, T6, CLEN
    s_or_b32 SLOTT, SLOTT, T0
    s_bitcmp1_b32 FLAGS, 7
    s_cbranch_scc1 .Lslot_fill_far                      // more than 64 literal trees: the descriptor from table memory
    v_readlane_b32 T0, VLHOFF, T6                       // the tree's descriptor
.Lslot_fill_have:
    s_bfm_b64 exec, 16, T1                              // the slot's lanes
    s_cmp_lt_i32 T0, 0
    s_cbranch_scc1 .Lslot_fill_single
    v_add_u32 VT0, T0, VLANE4
    ds_read_b32 VQL, VT0
    s_add_u32 T0, T0, SYMOFF
    s_waitcnt lgkmcnt(0)
    SPLIT_TREE VQL, VQB
    v_lshl_add_u32 VQB, VQB, 1, T0
    TO_COUNTS VQL, VT1
    s_mov_b64 exec, -1
    s_setpc_b64 LINKB
.Lslot_fill_single:                                     // a one-symbol tree: "matches" at length 0 (no bits), a two-entry list [x, x]
    s_and_b32 T0, T0, 0xffff                            // x = byte | context info << 8
    s_lshl_b32 CLEN, T0, 16
    s_or_b32 T0, T0, CLEN
    s_lshr_b32 CLEN, T1, 2
    s_add_u32 CLEN, CLEN, LDS_SPARE                     // 4 bytes per slot
    v_mov_b32 VT0, CLEN
    v_mov_b32 VT1, T0
    ds_write_b32 VT0, VT1
    v_mov_b32 VQB, CLEN
    v_mov_b32 VQL, 0
    v_writelane_b32 VQL, -1, T1                         // the slot's lane 0
    s_waitcnt lgkmcnt(0)
    s_mov_b64 exec, -1
    s_setpc_b64 LINKB
.Lslot_fill_far:                                        // (DESC_GATHER for one tree, the decisions on the scalar side)
    ds_read_b32 VT0, VZERO offset:LDS_MBW+20
    s_waitcnt lgkmcnt(0)
    v_add_u32 VT0, T6, VT0
    v_lshlrev_b32 VT0, 2, VT0
    ds_read_b32 VT0, VT0 offset:LDS_TM                  // h
    s_waitcnt lgkmcnt(0)
    v_lshlrev_b32 VT0, 2, VT0
    ds_read_b32 VT1, VT0 offset:LDS_TM+INFOOFF
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 T0, VT0
    v_readfirstlane_b32 CLEN, VT1
    s_add_u32 T0, T0, LDS_TM
    s_and_b32 vcc_lo, CLEN, 3
    s_cmp_eq_u32 vcc_lo, 1
    s_cbranch_scc0 .Lslot_fill_have
    s_lshr_b32 T0, CLEN, 16                             // a one-symbol tree
    s_bitset1_b32 T0, 31
    s_branch .Lslot_fill_have
// The context map row changed (a literal block switch): VT4 = tree index per context id -> VCMAP, VSLOT.  EXEC = all lanes.
.macro SLOT_REMAP
    v_lshlrev_b32 VCMAP, 1, VT4
    v_mov_b32 VSLOT, 0x80
    .irp sl, 0, 1, 2, 3
    s_bfe_u32 T2, SLOTT, 0x80000 + 8 * \sl
    s_lshl_b32 T2, T2, 1
    v_mov_b32 VT0, 16 * \sl
    v_cmp_eq_u32 vcc, T2, VCMAP
    v_cndmask_b32 VSLOT, VSLOT, VT0, vcc
    .endr
.endm
#endif
.Lent_cmh_far:
    s_bitset1_b32 FLAGS, 7
    DESC_GATHER VT4, VT4, 20, 3, VT0, VT1, vcc
    s_branch .Lent_cmh_have
.Lent_vdh_far:
    s_bitset1_b32 FLAGS, 8
    DESC_GATHER VDH4, VT3, 28, 7, VT0, VT1, vcc
    s_branch .Lent_vdh_have
.Lland:
    s_cmp_eq_u32 PFREE, 0
    s_cbranch_scc1 .Lland_chk
    LAND_BODY
.Lland_chk:
    s_cmp_ge_u32 POS, FLUSHAT
    s_cbranch_scc1 .Lland_flush
    s_setpc_b64 LINKB
.Lland_flush:
    FLUSH_BODY .Lland_flush_loop
    s_setpc_b64 LINKB
.Lland_ctx:
    s_cmp_lt_u32 POS, 2
    s_cbranch_scc1 .Lland_ctx_start
    s_cmp_eq_u32 PFREE, 0
    s_cbranch_scc1 .Lland_ctx_ring
    LAND_BODY
.Lland_ctx_ring:
    s_add_u32 T6, POS, SKEW
    v_bfe_u32 VPA, T6, 0, 11                            // ring address of the next literal (RMASK; a run never wraps: LIT_RUN_SETUP)
    v_add_u32 VT0, -1, VPA                              // (address arithmetic on the vector side: the scalar ALU is the
    v_add_u32 VT1, -2, VPA                              // unit all 16 waves of a CU share)
    v_and_b32 VT0, RMASK, VT0
    v_and_b32 VT1, RMASK, VT1
    ds_read_u8 VT0, VT0
    ds_read_u8 VT1, VT1
    s_waitcnt lgkmcnt(0)
.Lland_ctx_have:
    ds_read_u8 VT3, VT0 offset:LDS_ITAB                 // info(p1)
    ds_read_u8 VT4, VT1 offset:LDS_ITAB                 // info(p2)
    s_waitcnt lgkmcnt(0)
    v_and_b32 VB4, MB, VT3
    v_lshlrev_b32 VB4, SB, VB4
    v_and_b32 VT4, MB, VT4
    v_lshlrev_b32 VT4, SB, VT4
    v_and_or_b32 VC, VT3, MA, VT4
    ds_read_b32 VH, VC offset:LDS_CMH
    s_setpc_b64 LINKB

// start of the stream: the bytes before it count as 0 (src/lib.rs:389, 407); a copy is at least 2 bytes long, so nothing
// can be pending with fewer than 2 bytes of output
.Lland_ctx_start:
    s_add_u32 T6, POS, SKEW
    v_bfe_u32 VPA, T6, 0, 11
    v_mov_b32 VT0, 0
    v_mov_b32 VT1, 0
    s_cmp_eq_u32 POS, 0
    s_cbranch_scc1 .Lland_ctx_have
    s_and_b32 T7, SKEW, RMASK
    v_mov_b32 VT0, T7
    ds_read_u8 VT0, VT0                                 // the stream's first byte
    s_waitcnt lgkmcnt(0)
    s_branch .Lland_ctx_have

// Flush 1 KiB blocks of the ring to HBM (64 lanes x 16 B, both sides 16-byte aligned); pending bytes land first.
.Lflush:
    s_cmp_eq_u32 PFREE, 0
    s_cbranch_scc1 .Lflush_go
    LAND_BODY
.Lflush_go:
    FLUSH_BODY .Lflush_loop
    s_setpc_b64 LINKC

// Refill reached lane WLSTOP: either the staged chunk is used up (roll the two chunks, request the next one) or
// the cursor is within END_MARGIN dwords of the end of the stream (poison the block counters so that the loop leaves at
// its next R0 / R1 test; the C++ side finishes the stream with the exact end-of-input rules).
.Lspecial:
    s_cmp_lg_u32 WL, 64
    s_cbranch_scc1 .Lnear_end
    s_waitcnt vmcnt(0)
    s_mov_b64 exec, -1
    v_mov_b32 VCHA, VCHB
    s_add_u32 CBASE, CBASE, 64
    s_add_u32 T0, CBASE, 64
    v_add_u32 VT4, T0, VLANE
    v_min_u32 VT4, WENDM1, VT4
    v_lshlrev_b32 VT4, 2, VT4
    global_load_dword VCHB, VT4, INP
    s_mov_b64 exec, XLOOP
    s_mov_b32 WL, 0
    WIN_ROLLED
    // Issue priority by turns.  The arbiter of a SIMD serves its oldest wave first, and with the CU's scalar ALU saturated
    // (16 streams per CU) that starves the youngest waves: the oldest streams finish in 8.1 ms, the youngest in 11.0, and
    // a launch takes as long as its slowest wave.  Every 256 bytes of input a wave takes the next of the four priority
    // levels, offset by its slot in the SIMD, so that over a stream every wave spends the same time at every level.
    s_lshr_b32 T0, CBASE, 6
    s_add_u32 T0, T0, PRIOW
    s_and_b32 T0, T0, 3
    s_cmp_lt_u32 T0, 2
    s_cbranch_scc1 .Lprio_01
    s_cmp_eq_u32 T0, 2
    s_cbranch_scc1 .Lprio_2
    s_setprio 3
    s_branch .Lprio_done
.Lprio_2:
    s_setprio 2
    s_branch .Lprio_done
.Lprio_01:
    s_cmp_eq_u32 T0, 0
    s_cbranch_scc1 .Lprio_0
    s_setprio 1
    s_branch .Lprio_done
.Lprio_0:
    s_setprio 0
.Lprio_done:
    s_sub_u32 T0, WSAFE, CBASE
    s_cselect_b32 T0, 0, T0
    s_min_u32 WLSTOP, T0, 64
    s_cmp_lg_u32 WLSTOP, 0
    s_cbranch_scc1 .Lspecial_ret
.Lnear_end:
    s_mov_b32 WLSTOP, 64
    s_bitcmp1_b32 FLAGS, 0
    s_cbranch_scc1 .Lspecial_ret
    s_bitset1_b32 FLAGS, 0
    s_mov_b32 LBLEN_REAL, LBLEN
    s_mov_b32 IBLEN_REAL, IBLEN
    s_mov_b32 LBLEN, 0
    s_mov_b32 IBLEN, 0
.Lspecial_ret:
    s_setpc_b64 LINKA

    REFILL_STUB 1
    REFILL_STUB 2
    REFILL_STUB 3
    REFILL_STUB 4
    REFILL_STUB 5
    REFILL_STUB 6
    REFILL_STUB 8
#ifndef BRX_NO_SPEC
    REFILL_STUB 11
#endif
    REFILL_STUB 12
    REFILL_STUB 13
    REFILL_STUB 14
#ifdef BRX_DIST_RESIDENT
    REFILL_STUB 15
#endif

// ---- the uncommon copies.  Pending lanes used up: land and take the common path; otherwise the copy overlaps its
// source, is longer than 64 bytes or runs past the meta-block.
.Lcopy_slow_undo:
    s_sub_u32 PFREE, PFREE, CPY
    s_mov_b64 exec, XLOOP
.Lcopy_slow:
    s_sub_u32 MBLEFT, MBEND, POS
    s_min_u32 T0, DIST, 63
    s_min_u32 T0, T0, MBLEFT
    s_cmp_gt_u32 CPY, T0
    s_cbranch_scc1 .Lcopy_odd
    s_call_b64 LINKB, .Lland
    s_branch .Lcopy
.Lcopy_odd:
    s_cmp_lt_u32 DIST, CPY
    s_cbranch_scc0 .Lcopy_long                          // no overlap: 64..512 bytes (or past the meta-block: leaves)

// ---- a copy of <= 64 bytes that overlaps its source (distance < length): out[i] = src[i mod distance] (:1500-1503).
// The source bytes are final (anything pending is landed first), so this is a ring copy with a periodic lane index;
// lane mod distance by binary long division (lane < 64).  Longer copies go to .Lcopy_long or the C++ side.
.Lcopy_overlap:
    s_min_u32 T0, MBLEFT, 64
    s_cmp_gt_u32 CPY, T0
    s_cbranch_scc1 .Lcopy_long
    s_call_b64 LINKB, .Lland
    s_mov_b64 exec, -1
    s_sub_u32 T0, CPY, 1
    v_min_u32 VT2, T0, VLANE                            // switched-off lanes redo the last byte
    v_mov_b32 VT0, VT2
    .irp k, 5, 4, 3, 2, 1, 0
    s_lshl_b32 T2, DIST, \k
    v_subrev_u32 VT1, T2, VT0
    v_cmp_le_u32 vcc, T2, VT0
    v_cndmask_b32 VT0, VT0, VT1, vcc
    .endr
    s_sub_u32 T1, POS, DIST
    s_add_u32 T1, T1, SKEW
    v_add_u32 VT0, T1, VT0
    v_and_b32 VT0, RMASK, VT0
    ds_read_u8 VT3, VT0
    s_add_u32 T1, POS, SKEW
    v_add_u32 VT0, T1, VT2
    v_and_b32 VT0, RMASK, VT0
    s_waitcnt lgkmcnt(0)
    ds_write_b8 VT0, VT3
    s_mov_b64 exec, XLOOP
    s_add_u32 POS, POS, CPY
    s_mov_b32 PBASE, POS
    s_branch .Lcopy_tail

// min(T5, 64) bytes of a periodic copy at POS: byte i = period[(\off + i) mod DIST], period = the DIST < 64 final bytes at ring
// coordinate T3 (lane mod distance by binary long division, as .Lcopy_overlap).  EXEC = all lanes; the lanes beyond the count redo
// its last byte.  Moves POS and T5; clobbers T0 - T2, VT0 - VT3, vcc.
.macro PERIOD_CHUNK off
    s_min_u32 T0, T5, 64
    s_sub_u32 T1, T0, 1
    v_min_u32 VT2, T1, VLANE
    .if \off
    v_add_u32 VT0, \off, VT2
    .else
    v_mov_b32 VT0, VT2
    .endif
    .irp k, 6, 5, 4, 3, 2, 1, 0
    s_lshl_b32 T2, DIST, \k
    v_subrev_u32 VT1, T2, VT0
    v_cmp_le_u32 vcc, T2, VT0
    v_cndmask_b32 VT0, VT0, VT1, vcc
    .endr
    v_add_u32 VT0, T3, VT0
    v_and_b32 VT0, RMASK, VT0
    ds_read_u8 VT3, VT0
    s_add_u32 T1, POS, SKEW
    v_add_u32 VT0, T1, VT2
    v_and_b32 VT0, RMASK, VT0
    s_waitcnt lgkmcnt(0)
    ds_write_b8 VT0, VT3
    s_add_u32 POS, POS, T0
    s_sub_u32 T5, T5, T0
.endm
// ---- a copy of 65 bytes and more: 64-byte chunks, each read (ring or the stream's own HBM output), waited for and written before
// the next one (a distance below 64: see below).  Up to 512 bytes when the source is older than the ring -- every chunk then waits
// out a round trip to HBM, and from there on the C++ side's 16 B/lane steps win -- and up to COPY_NEAR_MAX when it is in the ring
// (round 6: a chunk is ~50 cycles there, a hand-over to the C++ side and back ~18 k: the break-even is far beyond the 8 KiB from
// which the C++ side has its periodic fills and direct far copies).  Anything longer goes to the C++ side.
#ifndef COPY_NEAR_MAX
#define COPY_NEAR_MAX 8191
#endif
.Lcopy_long:
    s_mov_b32 T0, COPY_NEAR_MAX
    s_cmp_gt_u32 DIST, RING
    s_cselect_b32 T0, 512, T0
    s_min_u32 T0, MBLEFT, T0
    s_cmp_gt_u32 CPY, T0
    s_cbranch_scc1 .Lx_r2
    s_call_b64 LINKB, .Lland
    s_mov_b32 T5, CPY                                   // bytes left
    s_cmp_lt_u32 DIST, 64
    s_cbranch_scc0 .Lcl_chunk
#ifdef BRX_NO_PERIOD_COPY
    s_branch .Lx_r2
#endif
    // Round 6: 65..512 bytes at a distance below 64 -- a run, a short period ("=====", zeros, a repeated record) -- used to leave for
    // the C++ side (one such copy is a third of what a 400-byte stream like monkey spends outside this loop).  The first 128 bytes
    // go as in .Lcopy_overlap, (offset + lane) mod distance from the final bytes in the ring; from then on ANY multiple of the
    // period between 64 and the bytes written so far serves as the distance, and the chunk loop below takes the rest:
    // DIST << (6 - floor(log2 DIST)) lies in [64, 128).  (Not after 64 bytes already: the smallest POWER-OF-TWO multiple can exceed
    // 64 + DIST -- 31 -> 124 -- and would read in front of the period.)  DIST itself is free here: the ring of last distances took
    // it at .Ldist_push_ok / gave it at .Ldist_zero, and nothing leaves through an exit that reports it before the next distance
    // is decoded.
    s_mov_b64 exec, -1
    s_sub_u32 T3, POS, DIST
    s_add_u32 T3, T3, SKEW                              // skewed ring coordinate of the period's first byte
    PERIOD_CHUNK 0
    PERIOD_CHUNK 64                                     // (CPY > 64: at least one byte)
    s_mov_b64 exec, XLOOP
    s_mov_b32 PBASE, POS
    s_flbit_i32_b32 T0, DIST                            // 31 - floor(log2 DIST)  (DIST >= 1)
    s_sub_u32 T0, T0, 25                                // 6 - floor(log2 DIST)
    s_lshl_b32 DIST, DIST, T0
    s_cmp_ge_u32 POS, FLUSHAT
    s_cbranch_scc0 .Lcp_noflush
    s_call_b64 LINKC, .Lflush
.Lcp_noflush:
    s_cmp_lg_u32 T5, 0
    s_cbranch_scc1 .Lcl_chunk
    s_branch .Lflush_back_cmd
.Lcl_chunk:
    s_mov_b64 exec, -1
    s_min_u32 T0, T5, 64
    s_sub_u32 T1, T0, 1
    v_min_u32 VT1, T1, VLANE                            // clamped lane
    s_sub_u32 T1, POS, DIST
    s_cmp_gt_u32 DIST, RING
    s_cbranch_scc1 .Lcl_far
    s_add_u32 T1, T1, SKEW
    v_add_u32 VT0, T1, VT1
    v_and_b32 VT0, RMASK, VT0
    ds_read_u8 VT2, VT0
    s_branch .Lcl_have
.Lcl_far:
    v_add_u32 VT0, T1, VT1
    buffer_load_ubyte VT2, VT0, RSRC, 0 offen
.Lcl_have:
    s_add_u32 T1, POS, SKEW
    v_add_u32 VT0, T1, VT1
    v_and_b32 VT0, RMASK, VT0
    s_waitcnt vmcnt(0) lgkmcnt(0)
    ds_write_b8 VT0, VT2
    s_mov_b64 exec, XLOOP
    s_add_u32 POS, POS, T0
    s_sub_u32 T5, T5, T0
    s_mov_b32 PBASE, POS
    s_cmp_ge_u32 POS, FLUSHAT
    s_cbranch_scc0 .Lcl_noflush
    s_call_b64 LINKC, .Lflush
.Lcl_noflush:
    s_cmp_lg_u32 T5, 0
    s_cbranch_scc1 .Lcl_chunk
    s_branch .Lflush_back_cmd

// ---- a transformed word (reference src/transformation/mod.rs, spec Appendix B): prefix + op(word) + suffix with
// op = identity / OmitFirstN / OmitLastN / UppercaseFirst / UppercaseAll.  Rare: everything pending lands first, the
// word is fetched and waited for, then prefix, middle and suffix go to the ring with three masked stores.
// In: T0 = byte offset of the word, T3 = transform id.
.Ldict_xform:
    s_cmp_gt_u32 T3, 120
    s_cbranch_scc1 .Lx_r2
    s_mul_i32 T4, T3, 20                                // sizeof(BrxTransform)
    s_load_dwordx4 s[12:15], XFP, T4                    // prefix[8], suffix[8]
    s_add_u32 T4, T4, 16
    s_load_dword s97, XFP, T4                           // plen | slen << 8 | op << 16
    s_waitcnt lgkmcnt(0)
    s_bfe_u32 T1, s97, 0x80010                          // op
    // (the landing -- it may flush, which takes T6 / T7 -- comes before the lengths are worked out: always for the two Uppercase
    // transforms, which need the word's bytes here; for the others only when 40 more lanes are not free)
    s_sub_u32 T4, T1, 1
    s_cmp_lt_u32 T4, 2
    s_cbranch_scc1 .Lxf_land
    s_cmp_gt_u32 PFREE, 23
    s_cbranch_scc0 .Lxf_landed
.Lxf_land:
    s_call_b64 LINKB, .Lland
.Lxf_landed:
    s_and_b32 T2, s97, 0xff                             // prefix length
    s_bfe_u32 T3, s97, 0x80008                          // suffix length
    s_mov_b32 T5, 0                                     // first word byte used
    s_mov_b32 T7, CPY                                   // word bytes used
    s_cmp_lt_u32 T1, 3
    s_cbranch_scc1 .Lxf_have                            // identity, UppercaseFirst, UppercaseAll: the whole word
    s_cmp_ge_u32 T1, 12
    s_cbranch_scc1 .Lxf_last
    s_sub_u32 T4, T1, 2                                 // OmitFirstN: word[min(N, len-1)..] (Q1)
    s_sub_u32 T6, CPY, 1
    s_min_u32 T5, T4, T6
    s_sub_u32 T7, CPY, T5
    s_branch .Lxf_have
.Lxf_last:
    s_sub_u32 T4, T1, 11                                // OmitLastN: word[..max(N, len) - N]
    s_max_u32 T7, CPY, T4
    s_sub_u32 T7, T7, T4
.Lxf_have:
    s_add_u32 CLEN, T2, T7
    s_add_u32 CLEN, CLEN, T3                            // transformed length (<= 40)
    s_cmp_gt_u32 CLEN, MBLEFT
    s_cbranch_scc1 .Lx_r2                               // :2105 on the transformed length (Q4): raised by the C++ side
    s_sub_u32 T4, T1, 1
    s_cmp_lt_u32 T4, 2
    s_cbranch_scc1 .Lxf_upper                           // UppercaseFirst / UppercaseAll: the word's bytes are needed HERE
    // (ONE transformed byte with nothing pending -- OmitFirst3 of a four-letter word -- would leave a single pending lane, and a literal
    // run takes its two context bytes from the LAST TWO pending lanes (LIT_CTX_ENTRY: "a copy is at least 2 bytes long"): that one is
    // stored straight into the ring, as the Uppercase forms are.  Found by the soak, tools/wide_fuzz.py 2 43 late.)
    s_add_u32 T4, PFREE, CLEN
    s_cmp_eq_u32 T4, 1
    s_cbranch_scc1 .Lxf_upper
    // ---- identity / OmitFirstN / OmitLastN (round 5: four transformed words in five of quality-5..11 text; libbrotlienc leans on
    // the dictionary, 40 % of the commands of a 4 KiB-window stream): prefix, the used part of the word and suffix JOIN THE PENDING
    // LANES -- prefix and suffix bytes from the transform's record (SGPRs), the word's bytes by a load that nothing waits for,
    // like an untransformed word (.Ldict_go).  Before, every such word landed everything pending and waited out its own load.
    s_mov_b64 exec, -1
    v_subrev_u32 VT1, PFREE, VLANE
    v_lshlrev_b32 VT1, 3, VT1                           // 8 * (lane - first lane of the prefix)
    s_bfm_b64 exec, T2, PFREE
    v_lshrrev_b64 v[54:55], VT1, s[12:13]
    v_mov_b32 VPEND, v54                                // prefix bytes
    s_add_u32 T4, PFREE, T2                             // first lane of the word's part
    s_add_u32 T0, T0, T5
    s_sub_u32 T0, T0, T4                                // + lane = byte offset in the dictionary
    s_bfm_b64 exec, T7, T4
    v_add_u32 VT0, T0, VLANE
    global_load_ubyte VPEND, VT0, DICTP
    s_add_u32 T4, T4, T7                                // first lane of the suffix
    s_mov_b64 exec, -1
    v_subrev_u32 VT1, T4, VLANE
    v_lshlrev_b32 VT1, 3, VT1
    s_bfm_b64 exec, T3, T4
    v_lshrrev_b64 v[54:55], VT1, s[14:15]
    v_mov_b32 VPEND, v54                                // suffix bytes
    s_mov_b64 exec, XLOOP
    s_add_u32 PFREE, PFREE, CLEN
    s_add_u32 POS, POS, CLEN
    s_branch .Lcopy_tail
.Lxf_upper:
    s_mov_b64 exec, -1
    // the used part of the word, one byte per lane (lanes past its end repeat the last byte; they are masked off below)
    s_add_u32 T0, T0, T5
    s_max_u32 T4, T7, 1
    s_sub_u32 T4, T4, 1
    v_min_u32 VT2, T4, VLANE
    v_add_u32 VT0, T0, VT2
    global_load_ubyte VT3, VT0, DICTP
    s_waitcnt vmcnt(0)
    s_sub_u32 T4, T1, 1
    s_cmp_lt_u32 T4, 2
    s_cbranch_scc0 .Lxf_store
    // UppercaseFirst / UppercaseAll (src/transformation/mod.rs:3-82): a-z ^ 32, the second byte of a 2-byte UTF-8
    // sequence ^ 32, the third of a 3-byte one ^ 5.  A word starting with 0x00 under UppercaseFirst makes the reference
    // panic (Q3): that one goes to the C++ side, which reports status 26.
    v_mov_b32 VT1, 0                                    // xor mask per lane
    s_mov_b32 T6, 0                                     // i
.Lxf_up_loop:
    v_readlane_b32 T4, VT3, T6                          // b = word[i]
    s_cmp_lt_u32 T4, 192
    s_cbranch_scc0 .Lxf_up_multi
    s_cmp_eq_u32 T1, 1                                  // UppercaseFirst on 0x00: reference panics
    s_cselect_b32 T0, 1, 0
    s_cmp_eq_u32 T4, 0
    s_cselect_b32 T0, T0, 0
    s_cmp_lg_u32 T0, 0
    s_cbranch_scc1 .Lx_r2
    s_sub_u32 T0, T4, 97
    s_cmp_le_u32 T0, 25
    s_cselect_b32 T0, 32, 0                             // a-z
    s_mov_b32 T4, T6                                    // flipped lane
    s_add_u32 T6, T6, 1
    s_branch .Lxf_up_mark
.Lxf_up_multi:
    s_cmp_lt_u32 T4, 224
    s_cselect_b32 T0, 32, 5
    s_cselect_b32 T4, 1, 2
    s_add_u32 T4, T6, T4                                // lane i + 1 (2-byte sequence) or i + 2 (3-byte)
    s_add_u32 T6, T4, 1
.Lxf_up_mark:
    v_mov_b32 VT0, T0
    v_cmp_eq_u32 vcc, T4, VLANE
    v_cndmask_b32 VT0, 0, VT0, vcc
    v_or_b32 VT1, VT1, VT0
    s_cmp_eq_u32 T1, 1
    s_cbranch_scc1 .Lxf_up_done                         // UppercaseFirst: one step
    s_cmp_lt_u32 T6, CPY
    s_cbranch_scc1 .Lxf_up_loop
.Lxf_up_done:
    v_xor_b32 VT3, VT3, VT1
.Lxf_store:
    v_lshlrev_b32 VT1, 3, VLANE
    v_lshrrev_b64 v[54:55], VT1, s[12:13]                 // lane i: prefix byte i
    s_add_u32 T4, POS, SKEW
    v_add_u32 VT0, T4, VLANE
    v_and_b32 VT0, RMASK, VT0
    s_bfm_b64 exec, T2, 0
    ds_write_b8 VT0, v54
    s_mov_b64 exec, -1
    s_add_u32 T4, T4, T2
    v_add_u32 VT0, T4, VLANE
    v_and_b32 VT0, RMASK, VT0
    s_bfm_b64 exec, T7, 0
    ds_write_b8 VT0, VT3
    s_mov_b64 exec, -1
    s_add_u32 T4, T4, T7
    v_lshrrev_b64 v[54:55], VT1, s[14:15]                 // lane i: suffix byte i
    v_add_u32 VT0, T4, VLANE
    v_and_b32 VT0, RMASK, VT0
    s_bfm_b64 exec, T3, 0
    ds_write_b8 VT0, v54
    s_mov_b64 exec, XLOOP
    s_add_u32 POS, POS, CLEN
    s_mov_b32 PBASE, POS
    s_branch .Lcopy_tail

// ======================================================================================================== block switches
// Block switch command (reference parse_block_switch_command + the block count that follows it, src/lib.rs:1226-1284,
// :957-987), inside the loop.  .Lswitch: in T7 = offset of the category's words in Lds::mbw (48 literals / 72 insert&copy /
// 96 distances: nbl, btype, btype_prev, blen, h_types, h_counts), return address LINKD.  Reads the block type code and the
// block count code, writes btype / btype_prev back to Lds::mbw and returns with SCC = 0, s13 = the new block type,
// T0 = the new count - 1.  Returns with SCC = 1 and NOTHING consumed when the counters are poisoned (the end of the input is
// near: the C++ side takes over) or one of the two codes is not a complete general code (one-symbol or incomplete codes:
// the C++ side reads them with the reference's exact rules).  Uses s12-s15, s36-s38, T2, T3, v14, v15, v20-v25, VLB;
// T4 / T5 (the literal context) and T7 are preserved.
.Lswitch:
    s_bitcmp1_b32 FLAGS, 0
    s_cbranch_scc1 .Lsw_ret                             // (SCC = 1)
    v_mov_b32 VT0, T7
    ds_read_b64 v[20:21], VT0 offset:LDS_MBW            // nbl, btype
    ds_read_b32 v22, VT0 offset:LDS_MBW+8               // btype_prev
    ds_read_b64 v[24:25], VT0 offset:LDS_MBW+16         // h_types, h_counts
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 s12, v20
    v_readfirstlane_b32 s13, v21
    v_readfirstlane_b32 s14, v22
    v_readfirstlane_b32 s36, v24
    v_readfirstlane_b32 s37, v25
    s_lshl_b32 s36, s36, 2
    s_lshl_b32 s37, s37, 2
    s_add_u32 s36, s36, LDS_TM                          // header of the block type code
    s_add_u32 s37, s37, LDS_TM                          // header of the block count code
    v_add_u32 VT1, s36, VLANE4
    v_add_u32 VT2, s37, VLANE4
    ds_read_b32 VLIM, VT1                               // header words of the type code
    ds_read_b32 v14, VT2                                // ... of the count code
    v_mov_b32 VT1, s36
    v_mov_b32 VT2, s37
    ds_read_b32 VT1, VT1 offset:INFOOFF                 // ... and their info words
    ds_read_b32 VT2, VT2 offset:INFOOFF
    s_waitcnt lgkmcnt(0)
    SPLIT_TREE VLIM, VBASE
    SPLIT_TREE v14, v15
    // both must be complete general codes: kind 2 in the info word, limit[15] = 2^15 (left-aligned: 0x80000000)
    v_readfirstlane_b32 T2, VT1
    v_readfirstlane_b32 T3, VT2
    s_and_b32 T2, T2, 3
    s_and_b32 T3, T3, 3
    s_cmp_lg_u32 T2, 2
    s_cbranch_scc1 .Lsw_ret
    s_cmp_lg_u32 T3, 2
    s_cbranch_scc1 .Lsw_ret
    v_readlane_b32 T2, VLIM, 15
    v_readlane_b32 T3, v14, 15
    s_cmp_lg_u32 T2, 0x80000000
    s_cbranch_scc1 .Lsw_ret
    s_cmp_lg_u32 T3, 0x80000000
    s_cbranch_scc1 .Lsw_ret
    LOOKUP2 VLIM, VBASE, s36, 1, ds_read_u16, SYMOFF, 12
    s_waitcnt lgkmcnt(0)
    v_readlane_b32 s15, VS, CLEN                        // block type code: 0 = the previous type, 1 = the next one, n = type n - 2
    s_add_u32 T2, s13, 1
    s_cmp_eq_u32 T2, s12
    s_cselect_b32 T2, 0, T2                             // (btype + 1) mod nbl
    s_sub_u32 T3, s15, 2
    s_cmp_eq_u32 s15, 1
    s_cselect_b32 T3, T2, T3
    s_cmp_eq_u32 s15, 0
    s_cselect_b32 T3, s14, T3
    v_mov_b32 v20, T3
    v_mov_b32 v21, s13
    v_mov_b32 VT0, T7
    ds_write_b32 VT0, v20 offset:LDS_MBW+4              // btype
    ds_write_b32 VT0, v21 offset:LDS_MBW+8              // btype_prev = the old type
    s_mov_b32 s13, T3
    LOOKUP2 v14, v15, s37, 1, ds_read_u16, SYMOFF, 13
    s_waitcnt lgkmcnt(0)
    v_readlane_b32 s15, VS, CLEN                        // block count code 0..25: base T3 and extra bits T2 (spec section 6)
    s_cmp_ge_u32 s15, 18
    s_cbranch_scc1 .Lsw_hi
    s_cmp_ge_u32 s15, 16
    s_cbranch_scc1 .Lsw_mid
    s_lshr_b32 T2, s15, 2                               // codes 0..15 in groups of four: 2 + g extra bits
    s_bfm_b32 T3, T2, 0
    s_lshl_b32 T3, T3, 4
    s_add_u32 T3, T3, 1                                 // 1 + 16 * (2^g - 1)
    s_add_u32 T2, T2, 2
    s_and_b32 s38, s15, 3
    s_lshl_b32 s38, s38, T2
    s_add_u32 T3, T3, s38
    s_branch .Lsw_extra
.Lsw_mid:                                               // 16, 17: 6 extra bits, bases 241 and 305
    s_mov_b32 T2, 6
    s_sub_u32 T3, s15, 16
    s_lshl_b32 T3, T3, 6
    s_add_u32 T3, T3, 241
    s_branch .Lsw_extra
.Lsw_hi:                                                // 18..24: code - 11 extra bits, base 241 + 2^(code - 11); 25: 24 bits
    s_sub_u32 T2, s15, 11
    s_bfm_b32 T3, 1, T2
    s_add_u32 T3, T3, 241
    s_cmp_eq_u32 s15, 25
    s_cselect_b32 T2, 24, T2
.Lsw_extra:
    TAKE_EXTRA s38, T3, T2, 14
    s_sub_u32 T0, s38, 1
    s_cmp_lg_u32 T0, T0                                 // SCC = 0
.Lsw_ret:
    s_setpc_b64 LINKD

// ---- insert&copy block switch: the tree of the new block type becomes the resident one
.Lx_r0_switch:
    s_mov_b32 T7, 72
    s_call_b64 LINKD, .Lswitch
    s_cbranch_scc1 .Lx_r0_bail
    s_mov_b32 IBLEN, T0
    // (the switch's own reads may have rolled the input staging into the last dwords of the stream and POISONED the
    // counters meanwhile -- .Lnear_end saved the stale ones: the new count is the real one, the live one stays poisoned)
    s_bitcmp1_b32 FLAGS, 0
    s_cselect_b32 IBLEN_REAL, IBLEN, IBLEN_REAL
    s_cselect_b32 IBLEN, 0, IBLEN
    ds_read_b32 v20, VZERO offset:LDS_MBW+24            // hi: handle table of the insert&copy trees
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 T2, v20
    s_add_u32 T2, T2, s13
    s_lshl_b32 T2, T2, 2
    v_mov_b32 VT0, T2
    ds_read_b32 VT1, VT0 offset:LDS_TM
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 T2, VT1
    s_lshl_b32 T2, T2, 2
    s_add_u32 T2, T2, LDS_TM
    s_add_u32 T3, T2, SYMOFF
    v_add_u32 VT0, T2, VLANE4
    ds_read_b32 VIACL, VT0
    s_waitcnt lgkmcnt(0)
    SPLIT_TREE VIACL, VIACB
    v_lshl_add_u32 VIACB, VIACB, 1, T3                  // folded bases (LOOKUP2F)
    TO_COUNTS VIACL, VT1
    s_branch .Lcmd_ticked
.Lx_r0_bail:                                            // insert&copy block count exhausted and not switched here (or poisoned)
    s_mov_b32 IBLEN, 0
    s_mov_b32 EXITC, 0
    s_branch .Lexit

// ---- literal block switch at a run's start (register-resident loops): the context -> tree map of the new block type
.Lsw_L:
    s_bitcmp1_b32 FLAGS, 4
    s_cbranch_scc1 .Lexit                               // block types that differ in context mode: the C++ side
    s_mov_b32 T7, 48
    s_call_b64 LINKD, .Lswitch
    s_cbranch_scc1 .Lexit                               // (nothing changed: the C++ side finds the run as it is)
    s_add_u32 LBLEN, T0, 1                              // (run form: literals that can be decoded before the next switch)
    s_bitcmp1_b32 FLAGS, 0                              // (poisoned during the switch: see .Lx_r0_switch)
    s_cselect_b32 LBLEN_REAL, LBLEN, LBLEN_REAL
    s_cselect_b32 LBLEN, 0, LBLEN
    s_and_b32 T2, FLAGS, 0x60
    s_cmp_eq_u32 T2, 0
    s_cbranch_scc1 .Lsw_L_done                          // one literal tree: no context map
    ds_read_b32 v20, VZERO offset:LDS_MBW+12            // cml: byte address of the literal context map
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 T2, v20
    s_lshl_b32 T3, s13, 6
    s_add_u32 T2, T2, T3
    s_mov_b64 exec, -1
    v_add_u32 VT0, T2, VLANE
    ds_read_u8 VT4, VT0 offset:LDS_TM                   // lane c: tree index of context id c
    s_waitcnt lgkmcnt(0)
#ifdef BRX_SLOTS
    s_bitcmp1_b32 FLAGS, 6
    s_cbranch_scc0 .Lsw_L_resident
    SLOT_REMAP
    s_bitcmp1_b32 FLAGS, 7                              // ... and the descriptor table of the per-literal loop (SL_RUN_END)
    s_cbranch_scc1 .Lsw_L_far
    v_lshlrev_b32 VT4, 2, VT4
    ds_bpermute_b32 VT4, VT4, VLHOFF
.Lsw_L_have:
    v_lshlrev_b32 VT3, 2, VLANE
    s_waitcnt lgkmcnt(0)
    ds_write_b32 VT3, VT4 offset:LDS_CMH
    s_mov_b64 exec, XLOOP
    s_setpc_b64 LINKB
.Lsw_L_far:
    DESC_GATHER VT4, VT4, 20, 3, VT0, VT1, vcc
    s_branch .Lsw_L_have
.Lsw_L_resident:
#endif
    v_lshlrev_b32 VCMAP, 1, VT4
    s_mov_b64 exec, XLOOP
.Lsw_L_done:
    s_setpc_b64 LINKB

// ---- literal block switch inside the per-literal loop (one context mode, trees in LDS): the context -> tree descriptor
// table of the new block type, and the descriptor of the literal in progress again
.Lx_lit_switch_u:
    s_mov_b32 T7, 48
    s_call_b64 LINKD, .Lswitch
    s_cbranch_scc1 .Lx_lit_switch
    s_mov_b32 LBLEN, T0
    s_bitcmp1_b32 FLAGS, 0                              // (poisoned during the switch: see .Lx_r0_switch)
    s_cselect_b32 LBLEN_REAL, LBLEN, LBLEN_REAL
    s_cselect_b32 LBLEN, 0, LBLEN
    ds_read_b32 v20, VZERO offset:LDS_MBW+12
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 T2, v20
    s_lshl_b32 T3, s13, 6
    s_add_u32 T2, T2, T3
    s_mov_b64 exec, -1
    v_add_u32 VT0, T2, VLANE
    ds_read_u8 VT4, VT0 offset:LDS_TM
    s_waitcnt lgkmcnt(0)
#ifdef BRX_SLOTS
    s_bitcmp1_b32 FLAGS, 6
    s_cbranch_scc0 .Lx_lsu_noslots
    SLOT_REMAP
.Lx_lsu_noslots:
#endif
    s_bitcmp1_b32 FLAGS, 7
    s_cbranch_scc1 .Lx_lsu_far
    v_lshlrev_b32 VT4, 2, VT4
    ds_bpermute_b32 VT4, VT4, VLHOFF
.Lx_lsu_have:
    v_lshlrev_b32 VT3, 2, VLANE
    s_waitcnt lgkmcnt(0)
    ds_write_b32 VT3, VT4 offset:LDS_CMH
    s_mov_b64 exec, XLOOP
    ds_read_b32 VH, VC offset:LDS_CMH
    s_branch .Llit_ticked_u
.Lx_lsu_far:
    DESC_GATHER VT4, VT4, 20, 3, VT0, VT1, vcc
    s_branch .Lx_lsu_have
.Lx_lit_switch_m:                                       // (block types of different context modes: the C++ side switches)
.Lx_lit_switch:                                         // literal block count exhausted (or poisoned), mid-run
    s_mov_b32 LBLEN, 0
    s_add_u32 INS, INS, 1                               // (the loop counter runs one behind)
    s_branch .Lexit

// ---- distance block switch: the trees of the new block type's four distance contexts, then this command's again
.Lx_dist_switch:
    s_mov_b32 T7, 96
    s_call_b64 LINKD, .Lswitch
    s_cbranch_scc1 .Lx_dist_bail
    s_mov_b32 DBLEN, T0
    ds_read_b32 v20, VZERO offset:LDS_MBW+16            // cmd: byte address of the distance context map (4 bytes per type)
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 T2, v20
    s_lshl_b32 T3, s13, 2
    s_add_u32 T2, T2, T3
    v_mov_b32 VT0, T2
    ds_read_b32 VT2, VT0 offset:LDS_TM
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 CMDW, VT2
    s_mov_b64 exec, -1
    v_lshrrev_b32 VT3, 1, VLANE                         // lane 2k (and 2k + 1) = context k
    v_lshlrev_b32 VT3, 3, VT3
    v_lshrrev_b32 VT3, VT3, CMDW
    v_and_b32 VT3, 0xff, VT3
    s_bitcmp1_b32 FLAGS, 8
    s_cbranch_scc1 .Lx_ds_far
    v_lshlrev_b32 VT3, 2, VT3
    ds_bpermute_b32 VDH4, VT3, VDHOFF
.Lx_ds_have:
    s_waitcnt lgkmcnt(0)
    s_mov_b64 exec, XLOOP
    s_mov_b32 T2, 0xc0000000
    v_writelane_b32 VDH4, T2, 8                         // "tree" of an implicit distance code 0
    s_nop 0                                             // (a VALU-written VGPR needs one wait state before v_readlane)
#ifdef BRX_DIST_RESIDENT
    s_call_b64 LINKB, .Lload_dtrees
    v_readlane_b32 DTREE, VDH4, DCTX
#else
    v_readlane_b32 DTREE, VDH4, DCTX
    s_nop 1
    v_add_u32 VT0, DTREE, VLANE4
    ds_read_b32 VDHV, VT0
#endif
    s_cmp_lt_i32 DTREE, 0
    s_cbranch_scc0 .Ldist_ticked
    s_and_b32 DCODE, DTREE, 0xffff                      // a one-symbol tree (always a last-distance code here)
    s_branch .Ldist_ring_s
.Lx_ds_far:
    DESC_GATHER VDH4, VT3, 28, 7, VT0, VT1, s[86:87]    // (T2 / T3 as the mask: vcc may hold the early insert&copy compare)
    s_branch .Lx_ds_have
.Lx_dist_bail:
    s_mov_b32 DBLEN, 0
    s_mov_b32 INS, 0
    s_branch .Lexit
#ifdef BRX_DIST_RESIDENT
// limits and folded bases ((base << 2) + address of the symbol list) of the four distance-context trees named by VDH4 into
// v104..v111.  A one-symbol tree (descriptor < 0) has no header: its pair is never used (.Ldist_special).  Clobbers T2, T3, VT0, VLB.
.Lload_dtrees:
    s_mov_b32 T3, 0
    s_mov_b32 DSPEC, 0x01000115
.Lld_loop:
    v_readlane_b32 T2, VDH4, T3
    s_cmp_lt_i32 T2, 0
    s_cbranch_scc1 .Lld_single
    v_add_u32 VT0, T2, VLANE4
    ds_read_b32 VLIM, VT0
    s_add_u32 T2, T2, SYMOFF
    s_waitcnt lgkmcnt(0)
    SPLIT_TREE VLIM, VBASE
    v_lshl_add_u32 VBASE, VBASE, 2, T2
    TO_COUNTS VLIM, VT0
    s_set_gpr_idx_on T3, 8                              // VGPR index mode, destination + T3
    v_mov_b32 VDTREES, VLIM
    v_mov_b32 VDTREES1, VBASE
    s_set_gpr_idx_off
.Lld_next:
    s_add_u32 T3, T3, 2
    s_cmp_lt_u32 T3, 8
    s_cbranch_scc1 .Lld_loop
    s_setpc_b64 LINKB
.Lld_single:
    s_bitset1_b32 DSPEC, T3
    s_add_u32 T2, T3, 16
    s_bitset1_b32 DSPEC, T2
    s_branch .Lld_next
#endif

// ======================================================================================================== exits
.Lx_dist_unfit:                                         // back to R1 with the literals done: the C++ side reads the distance
    s_add_u32 SNAV, SNAV, CLEN                          // (un-take the symbol: the bit cursor is derived from SNAV)
    s_add_u32 DBLEN, DBLEN, 1
    s_mov_b32 INS, 0
    s_branch .Lexit
.Lx_dist_bad:                                           // non-positive distance: raised by the C++ side at R2
    s_mov_b32 DIST, 0
    s_mov_b32 EXITC, 3
    s_branch .Lexit
.Lx_r2:
    s_mov_b32 EXITC, 2
.Lexit:
    s_mov_b64 exec, XLOOP
    s_call_b64 LINKB, .Lland
    s_waitcnt vmcnt(0) lgkmcnt(0)
#ifdef BRX_PROF
    s_memtime s[14:15]
    s_waitcnt lgkmcnt(0)
    s_sub_u32 s14, s14, s12                             // cycles inside the loop (low word)
    v_mov_b32 VT0, s20
    v_mov_b32 VT1, s21
    v_mov_b32 VT2, s22
    v_mov_b32 VT3, s14
    ds_add_u32 VZERO, VT0 offset:LDS_PAD+40             // pad[10]: cycles waiting for vmcnt at landings
    ds_add_u32 VZERO, VT1 offset:LDS_PAD+44             // pad[11]: landings with something pending
    ds_add_u32 VZERO, VT2 offset:LDS_PAD+48             // pad[12]: bytes landed
    ds_add_u32 VZERO, VT3 offset:LDS_PAD+52             // pad[13]: cycles inside the assembly loop
    v_mov_b32 VT0, s23
    v_mov_b32 VT1, s29
    v_mov_b32 VT2, s30
    v_mov_b32 VT3, s31
    ds_add_u32 VZERO, VT0 offset:LDS_PAD+24             // pad[6]: insert&copy symbol sections
    ds_add_u32 VZERO, VT1 offset:LDS_PAD+28             // pad[7]: literal sections
    ds_add_u32 VZERO, VT2 offset:LDS_PAD+56             // pad[14]: distance sections
    ds_add_u32 VZERO, VT3 offset:LDS_PAD+60             // pad[15]: copy + tail sections
#endif
    // real block counters if they were poisoned
    s_bitcmp1_b32 FLAGS, 0
    s_cselect_b32 LBLEN, LBLEN_REAL, LBLEN
    s_cselect_b32 IBLEN, IBLEN_REAL, IBLEN
    // bit cursor: 32 * (CBASE + WL) - NAV
    s_add_u32 T0, CBASE, WL
    s_lshl_b32 T0, T0, 5
    s_sub_u32 T0, T0, SNAV
    s_sub_u32 T0, T0, 32                                // (SNAV = valid bits - 32)
    v_mov_b32 VT0, T0
    v_mov_b32 VT1, 0
    ds_write_b32 VZERO, VT0 offset:LDS_ST+12            // bitpos (st[3], st[4])
    ds_write_b32 VZERO, VT1 offset:LDS_ST+16
    v_mov_b32 VT0, POS
    ds_write_b32 VZERO, VT0 offset:LDS_ST+40
    v_mov_b32 VT0, VFL
    ds_write_b32 VZERO, VT0 offset:LDS_ST+48
    v_lshlrev_b32 VT0, 2, VLANE
    s_mov_b64 exec, 15
    ds_write_b32 VT0, VRING offset:LDS_ST+56
    s_mov_b64 exec, XLOOP
    v_mov_b32 VT0, LBLEN
    ds_write_b32 VZERO, VT0 offset:LDS_MBW+60
    v_mov_b32 VT0, IBLEN
    ds_write_b32 VZERO, VT0 offset:LDS_MBW+84
    v_mov_b32 VT0, DBLEN
    ds_write_b32 VZERO, VT0 offset:LDS_MBW+108
    s_cmp_eq_u32 DCTX, 8
    s_cselect_b32 IZ, 1, 0
    s_sub_u32 MBLEFT, MBEND, POS
    v_mov_b32 v20, MBLEFT
    v_mov_b32 v21, INS
    v_mov_b32 v22, CPY
    v_mov_b32 v23, IZ
    ds_write_b128 VZERO, v[20:23] offset:LDS_MBW+128
    s_cmp_eq_u32 EXITC, 3
    s_cselect_b32 T0, 1, 0
    s_cselect_b32 EXITC, 2, EXITC
    // bit 4 of the exit word: the cursor is within the last dwords of the stream (the loop will not run again): the C++ side
    // then finishes the meta-block in ONE call instead of one command per call with a futile re-entry here in between
    s_add_u32 T1, CBASE, WL
    s_add_u32 T1, T1, 3
    s_cmp_ge_u32 T1, WSAFE
    s_cselect_b32 T1, 16, 0
    s_or_b32 EXITC, EXITC, T1
    v_mov_b32 v20, DIST
    v_mov_b32 v21, T0
    v_mov_b32 v22, EXITC
    ds_write_b64 VZERO, v[20:21] offset:LDS_MBW+144     // distance, distance-is-bad
    ds_write_b32 VZERO, v22 offset:LDS_MBW+152          // exit point
    s_waitcnt lgkmcnt(0)
    s_mov_b64 exec, -1                                  // the C++ segments run with all lanes
