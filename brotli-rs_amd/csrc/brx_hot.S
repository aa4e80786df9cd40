// brx_hot.S -- the command loop of one compressed meta-block, hand-written for gfx950 (CDNA4).
//
// Reference states DataMetaBlockBegin .. CopyLiterals (src/lib.rs:2003-2141): insert&copy symbol, extra bits,
// context-modelled literals, distance symbol / last-distance ring, window copy, dictionary word.  One wavefront = one
// stream; all decoder state is wave-uniform (SGPRs for what steers control flow, "uniform VGPRs" -- the same value in
// every lane -- for the bit window and the arithmetic, see TAKE below); the 64 lanes are used as (a) a 256-byte staging
// buffer of the compressed input (v_readlane feeds the 64-bit bit window), (b) comparators of the canonical prefix-code
// lookup (lane L holds the left-aligned exclusive upper bound of the length-L codes: one v_cmp + s_ff1 = code length),
// (c) byte movers of a copy (one byte per lane, LDS ring or buffer_load of the stream's own HBM output).
//
// This file is preprocessed (register names) and pasted into ONE asm statement of brx_kernels.hip
// (asm_commands()).  It talks to the C++ segments only through the parked state in LDS (Lds::st, Lds::mbw):
//   entry : always at resume point R1 (insert_len / copy_len / implicit_zero of the current command known)
//   exit  : mbw[MBW_EXIT] = 0 (R0: insert&copy symbol due), 1 (R1), 2 (R2: distance of the current command known).
// Everything unusual leaves through one of those points and is handled by generic_commands() in C++, which runs one
// command and hands back: block switches, copies longer than 512 bytes (or long and closer than 64 bytes), any error (the C++ side re-decodes and raises it), the last 256 bits of the stream
// (so no end-of-input test is needed here: every bit consumed below is a real bit), a ragged first flush block.
//
// Preconditions (the HC_START call of generic_commands() sets mbw[MBW_ASM]): every literal and
// distance tree is a complete general code or a one-symbol code, every insert&copy tree a complete general code, all
// of them and the context maps resident in LDS table memory, <= 64 literal and <= 64 distance trees, input < 2^28
// bytes, pos + MLEN <= capacity.  (A ragged flush cursor at entry hands straight back to the C++ side.)

#define LDS_TM 4096
#define LDS_ST 9728
#define LDS_MBW 9920
#define RMASK 4095
// the code-length scratch area (Lds::lens, 768 B) is free during the command loop: context tables live there
#define LDS_ATAB 8960
#define LDS_BTAB 9216
#define LDS_CMH 9472

// ---- SGPRs (s36-s38: scratch during entry)
#define WL s39
#define MAXA s40
#define WLSTOP s41
#define INP s[42:43]
#define WENDM1 s44
#define WSAFE s45
#define CBASE s46
#define DIST s47
#define RSRC s[48:51]
#define POS s52
#define SKEW s53
#define VFL s54
#define FLUSHAT s55
#define WINDOW s56
#define D0 s57
#define D1 s58
#define D2 s59
#define D3 s60
#define MBLEFT s61
#define INS s62
#define CPY s63
#define IZ s64
#define LBLEN s65
#define IBLEN s66
#define DBLEN s67
#define P1 s68
#define PENDN s70
#define HISYM s71
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
#define LINKB s[98:99]
#define LINKC s[100:101]
#define LINKD s[24:25]
#define NPOST s4
#define NDIRECT s5
#define POSTMASK s6
#define NDIRECT1 s7
#define NDIR16 s8
#define NPOST1 s9
#define DTREE s10
#define XFP s[26:27]
#define PENDEND s28
// ---- VGPRs
#define VZERO v0
#define VLANE v1
#define VLANE4 v2
#define VLANE16 v3
#define VCHA v4
#define VCHB v5
#define VVA v6
#define VVB v7
#define VLHOFF v9
#define VDHOFF v10
#define VPEND v12
#define VDICTINFO v13
#define VQ v[14:17]
#define VT0 v18
#define VT1 v19
#define VT2 v20
#define VT3 v21
#define VT4 v22
#define VPENB v23
#define VR v24
#define VU v25
#define VBASE v26
#define VI v27
#define VIACL v66
#define VIACB v67
#define VE v30
#define VLIM v31
#define VWIN v[32:33]
#define VWINLO v32
#define VWINHI v33
#define VNAV v34
#define VRF v[36:37]
#define VRFLO v36
#define VRFHI v37
// (v40-v47 are callee-saved in the AMDGPU calling convention: using them would make the wrapper spill them to scratch on
// every call)
#define VA1 v38
#define VB1 v39
#define VB2 v49
#define VC v50
#define VH v51
#define VX v52
#define VN v53
#define VHH v54
#define VLC v55
#define VDH4 v64
#define VDHV v48
#define VDHB v65
#define VCLA v68
#define VCLB v69

// The bit window lives in a VGPR pair (the same value in every lane) and is worked on by the VECTOR ALU: the scalar
// ALU issues one instruction per SIMD every 4 cycles and is the bottleneck of this loop (profiles/r01g_pmc.csv), the
// vector ALU is mostly idle.  Bits are taken from the low end of VWIN; VNAV = number of valid bits (>= 32 after a
// REFILL_CHECK).  Only values that steer control flow or index lanes are moved to SGPRs (v_readfirstlane).
.macro TAKE n
    v_lshrrev_b64 VWIN, \n, VWIN
    v_subrev_u32 VNAV, \n, VNAV
.endm
.macro REFILL_CHECK id
    v_cmp_gt_u32 vcc, 32, VNAV
    s_cbranch_vccnz .Lrf_stub_\id
.Lrf_back_\id:
.endm
// Out-of-line part of a refill: next dword of the staged input (lane WL of chunk A) enters the window.
.macro REFILL_STUB id
.Lrf_stub_\id:
    v_readlane_b32 T0, VCHA, WL
    v_mov_b32 VRFHI, 0
    s_nop 1                                             // gfx940+: VALU-written SGPR read by a VALU: 2 wait states
    v_mov_b32 VRFLO, T0
    v_lshlrev_b64 VRF, VNAV, VRF
    v_or_b32 VWINLO, VWINLO, VRFLO
    v_or_b32 VWINHI, VWINHI, VRFHI
    v_add_u32 VNAV, 32, VNAV
    s_add_u32 WL, WL, 1
    s_cmp_lg_u32 WL, WLSTOP
    s_cbranch_scc1 .Lrf_back_\id
    s_call_b64 LINKA, .Lspecial
    s_branch .Lrf_back_\id
.endm
// Canonical prefix-code lookup (table layout: brx_kernels.hip, "Table layout in table memory").
// lim = per-lane limit[L] << 16 (lane 0: 0), base = per-lane base[L] of the tree (lane L, L = 1..15).
// Out: CLEN = code length (SGPR), VI = index into the tree's sorted symbol list (VGPR).  Clobbers T2, T3, VR, VU, vcc.
// Every code is complete (precondition), so some lane always matches; the lowest matching lane is the length.
.macro LOOKUP lim, base
    v_bfrev_b32 VR, VWINLO
    v_lshrrev_b32 VU, 1, VR
    v_cmp_lt_u32 vcc, VU, \lim
    s_ff1_i32_b32 CLEN, vcc_lo
    v_readlane_b32 T3, \base, CLEN
    s_sub_u32 T2, 32, CLEN
    v_lshrrev_b32 VI, T2, VR                            // (two instructions between the v_readlane and its VALU reader)
    v_add_u32 VI, T3, VI
.endm

// ======================================================================================================== entry
    s_waitcnt vmcnt(0) lgkmcnt(0)
    v_mov_b32 VZERO, 0
    v_mbcnt_lo_u32_b32 VLANE, -1, 0
    v_mbcnt_hi_u32_b32 VLANE, -1, VLANE
    v_and_b32 VLANE4, 15, VLANE
    v_lshlrev_b32 VLANE4, 2, VLANE4
    v_lshlrev_b32 VLANE16, 4, VLANE
    // parked decoder state: st[0..17], st[23..28], st[36..37]
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
    v_readfirstlane_b32 D0, v34
    v_readfirstlane_b32 D1, v35
    ds_read_b64 v[20:21], VZERO offset:LDS_ST+64        // dist2, dist3
    ds_read_b32 v22, VZERO offset:LDS_ST+92             // t_dict (st[23], st[24]: only 4-byte aligned)
    ds_read_b32 v23, VZERO offset:LDS_ST+96
    ds_read_b32 v26, VZERO offset:LDS_ST+100            // t_xforms (st[25], st[26])
    ds_read_b32 v27, VZERO offset:LDS_ST+104
    ds_read_b32 v24, VZERO offset:LDS_ST+108            // t_lut (st[27], st[28])
    ds_read_b32 v25, VZERO offset:LDS_ST+112
    ds_read_b64 v[28:29], VZERO offset:LDS_ST+144       // insert&copy / dictionary info table
    s_and_b32 s49, s49, 0xffff
    s_mov_b32 s51, 0x00020000
    s_sub_u32 WENDM1, T0, 1
    s_lshr_b32 WSAFE, T2, 5
    s_sub_u32 WSAFE, WSAFE, 8
    s_cselect_b32 WSAFE, 0, WSAFE                       // borrow -> 0
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
    v_readfirstlane_b32 D2, v20
    v_readfirstlane_b32 D3, v21
    v_readfirstlane_b32 s76, v22
    v_readfirstlane_b32 s77, v23
    v_readfirstlane_b32 T4, v24
    v_readfirstlane_b32 T5, v25
    v_readfirstlane_b32 s26, v26
    v_readfirstlane_b32 s27, v27
    v_readfirstlane_b32 s74, v28
    v_readfirstlane_b32 s75, v29
    // context LUTs (3 x 64 dwords) and the dictionary info vector (dwords 1408.. of the insert&copy table)
    v_lshlrev_b32 VT0, 2, VLANE
    s_nop 4                                             // v_readfirstlane -> VMEM address SGPR: 5 wait states
    global_load_dword v28, VT0, s[88:89]                // Lut0
    global_load_dword v29, VT0, s[88:89] offset:256     // Lut1
    global_load_dword v30, VT0, s[88:89] offset:512     // Lut2
    v_add_u32 VT1, 11264, VT0
    global_load_dword VDICTINFO, VT1, IACTAB
    // meta-block words
    ds_read_b128 v[20:23], VZERO offset:LDS_MBW+0       // npostfix, ndirect, cmode_w, cml
    ds_read_b128 v[24:27], VZERO offset:LDS_MBW+16      // cmd, hl, hi, hd
    ds_read_b64 v[32:33], VZERO offset:LDS_MBW+32       // ntl, ntd
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 NPOST, v20
    v_readfirstlane_b32 NDIRECT, v21
    v_readfirstlane_b32 T0, v22                         // cmode_w
    v_readfirstlane_b32 T1, v23                         // cml
    v_readfirstlane_b32 T2, v24                         // cmd
    v_readfirstlane_b32 T6, v25                         // hl
    v_readfirstlane_b32 T7, v26                         // hi
    v_readfirstlane_b32 s92, v27                        // hd
    v_readfirstlane_b32 s93, v32                        // ntl
    v_readfirstlane_b32 s94, v33                        // ntd
    ds_read_b32 v20, VZERO offset:LDS_MBW+52            // L.btype
    ds_read_b32 v21, VZERO offset:LDS_MBW+60            // L.blen
    ds_read_b32 v22, VZERO offset:LDS_MBW+76            // I.btype
    ds_read_b32 v23, VZERO offset:LDS_MBW+84            // I.blen
    ds_read_b32 v24, VZERO offset:LDS_MBW+100           // D.btype
    ds_read_b32 v25, VZERO offset:LDS_MBW+108           // D.blen
    ds_read_b128 v[32:35], VZERO offset:LDS_MBW+128     // mb_left, insert_len, copy_len, implicit_zero
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 s95, v20                        // L.btype
    v_readfirstlane_b32 LBLEN, v21
    v_readfirstlane_b32 s97, v22                        // I.btype
    v_readfirstlane_b32 IBLEN, v23
    v_readfirstlane_b32 DCODE, v24                      // D.btype (DCODE is free until the first distance)
    v_readfirstlane_b32 DBLEN, v25
    v_readfirstlane_b32 MBLEFT, v32
    v_readfirstlane_b32 INS, v33
    v_readfirstlane_b32 CPY, v34
    v_readfirstlane_b32 IZ, v35
    // per-tree descriptors: lane t -> LDS byte address of tree t's header (tm word h -> LDS_TM + 4h), or for a
    // one-symbol tree (kind 1, zero-bit code, SURVEY Q5) 0x80000000 | symbol
    s_sub_u32 s93, s93, 1
    v_min_u32 VT0, s93, VLANE
    v_add_u32 VT0, T6, VT0
    v_lshlrev_b32 VT0, 2, VT0
    ds_read_b32 VLHOFF, VT0 offset:LDS_TM
    s_sub_u32 s94, s94, 1
    v_min_u32 VT0, s94, VLANE
    v_add_u32 VT0, s92, VT0
    v_lshlrev_b32 VT0, 2, VT0
    ds_read_b32 VDHOFF, VT0 offset:LDS_TM
    s_waitcnt lgkmcnt(0)
    v_lshlrev_b32 VLHOFF, 2, VLHOFF
    v_lshlrev_b32 VDHOFF, 2, VDHOFF
    ds_read_b32 VT0, VLHOFF offset:LDS_TM+64            // header word 16: kind | max_len << 8 | symbol << 16
    ds_read_b32 VT1, VDHOFF offset:LDS_TM+64
    v_add_u32 VLHOFF, LDS_TM, VLHOFF
    v_add_u32 VDHOFF, LDS_TM, VDHOFF
    s_waitcnt lgkmcnt(0)
    v_and_b32 VT2, 3, VT0
    v_lshrrev_b32 VT0, 16, VT0
    v_or_b32 VT0, 0x80000000, VT0
    v_cmp_eq_u32 vcc, 1, VT2
    v_cndmask_b32 VLHOFF, VLHOFF, VT0, vcc
    v_and_b32 VT2, 3, VT1
    v_lshrrev_b32 VT1, 16, VT1
    v_or_b32 VT1, 0x80000000, VT1
    v_cmp_eq_u32 vcc, 1, VT2
    v_cndmask_b32 VDHOFF, VDHOFF, VT1, vcc
    // insert&copy tree of the current block type: header words stay resident
    s_add_u32 T7, T7, s97
    s_lshl_b32 T7, T7, 2
    v_mov_b32 VT0, T7
    ds_read_b32 VT1, VT0 offset:LDS_TM
    // literal context map row of the current block type (64 bytes, lanes 0..15), distance map word, context mode
    s_lshl_b32 T6, s95, 6
    s_add_u32 T1, T1, T6
    v_add_u32 VT0, T1, VLANE
    ds_read_u8 VT4, VT0 offset:LDS_TM                   // lane c: tree index of context id c
    s_lshl_b32 T6, DCODE, 2
    s_add_u32 T2, T2, T6
    v_mov_b32 VT0, T2
    ds_read_b32 VT2, VT0 offset:LDS_TM
    s_lshl_b32 T0, T0, 2
    s_add_u32 T0, T0, s95
    v_mov_b32 VT0, T0
    ds_read_u8 VT3, VT0 offset:LDS_TM
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 T7, VT1                         // h of the insert&copy tree
    v_readfirstlane_b32 CMDW, VT2
    v_readfirstlane_b32 T0, VT3                         // context mode
    // CMH[c] = descriptor of the literal tree of context id c; VDH4 lane k = descriptor of the distance tree of
    // distance context k (both for the current block types; a block switch leaves the loop and re-enters here)
    v_lshlrev_b32 VT4, 2, VT4
    ds_bpermute_b32 VT4, VT4, VLHOFF
    v_lshlrev_b32 VT3, 3, VLANE
    v_lshrrev_b32 VT3, VT3, CMDW
    v_and_b32 VT3, 0xff, VT3
    v_lshlrev_b32 VT3, 2, VT3
    ds_bpermute_b32 VDH4, VT3, VDHOFF
    v_lshlrev_b32 VT3, 2, VLANE
    s_waitcnt lgkmcnt(0)
    ds_write_b32 VT3, VT4 offset:LDS_CMH
    s_lshl_b32 T7, T7, 2
    s_add_u32 T7, T7, LDS_TM
    s_add_u32 HISYM, T7, 128
    v_add_u32 VT0, T7, VLANE4
    ds_read_b32 VIACL, VT0                              // limits and bases of the insert&copy tree stay resident
    ds_read_b32 VIACB, VT0 offset:64
    // context vectors (see ctx_vectors() in brx_kernels.hip): id = (A[p1] | B[p2]) & 63
    s_waitcnt vmcnt(0)
    v_lshlrev_b32 VT0, 2, VLANE
    v_and_b32 VT0, 0x3f, VT0
    s_mov_b32 T6, 0x01010101
    v_mul_lo_u32 VT0, VT0, T6
    v_add_u32 VVA, 0x03020100, VT0                      // mode 0: A = p & 63
    v_mov_b32 VVB, 0
    s_cmp_eq_u32 T0, 1
    s_cbranch_scc0 .Lcm_not1
    v_mul_lo_u32 VVA, VLANE, T6                         // mode 1: A = p >> 2
.Lcm_not1:
    s_cmp_eq_u32 T0, 2
    s_cbranch_scc0 .Lcm_not2
    v_mov_b32 VVA, v28                                  // mode 2: A = Lut0, B = Lut1
    v_mov_b32 VVB, v29
.Lcm_not2:
    s_cmp_eq_u32 T0, 3
    s_cbranch_scc0 .Lcm_not3
    v_and_b32 VVA, 0x1f1f1f1f, v30                      // mode 3: A = Lut2 << 3, B = Lut2
    v_lshlrev_b32 VVA, 3, VVA
    v_mov_b32 VVB, v30
.Lcm_not3:
    // the tables hold (x & 63) << 2: A[p1] | B[p2] is then the byte offset of the context's entry in CMH
    v_and_b32 VVA, 0x3f3f3f3f, VVA
    v_and_b32 VVB, 0x3f3f3f3f, VVB
    v_lshlrev_b32 VVA, 2, VVA
    v_lshlrev_b32 VVB, 2, VVB
    v_lshlrev_b32 VT0, 2, VLANE
    ds_write_b32 VT0, VVA offset:LDS_ATAB
    ds_write_b32 VT0, VVB offset:LDS_BTAB
    // bit window
    v_readlane_b32 s36, VCHA, 0
    v_readlane_b32 s37, VCHA, 1
    s_lshr_b64 s[36:37], s[36:37], T3
    s_sub_u32 s38, 64, T3
    v_mov_b32 VWINLO, s36
    v_mov_b32 VWINHI, s37
    v_mov_b32 VNAV, s38
    s_mov_b32 WL, 2
    s_sub_u32 T0, WSAFE, CBASE
    s_cselect_b32 T0, 0, T0
    s_min_u32 WLSTOP, T0, 64
    s_bfm_b32 POSTMASK, NPOST, 0
    s_add_u32 NDIRECT1, NDIRECT, 1
    s_add_u32 NDIR16, NDIRECT, 16
    s_add_u32 NPOST1, NPOST, 1
    s_mov_b32 PENDN, 0
    s_mov_b32 FLAGS, 0
    s_mov_b32 EXITC, 1
    s_and_b32 T0, VFL, 0xfffffc00
    s_add_u32 T0, T0, 2048
    s_sub_u32 FLUSHAT, T0, SKEW
    // literal context of the first literal: last two bytes of the output
    s_add_u32 T0, POS, SKEW
    s_sub_u32 T1, T0, 1
    s_and_b32 T1, T1, RMASK
    v_mov_b32 VT0, T1
    ds_read_u8 VT1, VT0
    s_sub_u32 T1, T0, 2
    s_and_b32 T1, T1, RMASK
    v_mov_b32 VT0, T1
    ds_read_u8 VT2, VT0
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 P1, VT1
    v_readfirstlane_b32 T5, VT2
    s_cmp_ge_u32 POS, 1
    s_cselect_b32 P1, P1, 0
    s_cmp_ge_u32 POS, 2
    s_cselect_b32 T5, T5, 0
    v_mov_b32 VT0, P1
    v_mov_b32 VT3, T5
    ds_read_u8 VA1, VT0 offset:LDS_ATAB                 // A[p1]
    ds_read_u8 VB1, VT0 offset:LDS_BTAB                 // B[p1] (becomes B[p2] after the next literal)
    ds_read_u8 VB2, VT3 offset:LDS_BTAB                 // B[p2]
    // not enough input left for the fast loop, or ragged flush cursor: hand straight back
    s_cmp_lt_u32 WLSTOP, 3
    s_cbranch_scc1 .Lexit
    s_and_b32 T0, VFL, 1023
    s_cmp_lg_u32 T0, 0
    s_cbranch_scc1 .Lexit
    s_branch .Lr1

// ======================================================================================================== R0
.Lcmd:
    s_sub_u32 IBLEN, IBLEN, 1
    s_cbranch_scc1 .Lx_r0_switch
    LOOKUP VIACL, VIACB
    v_lshl_add_u32 VT0, VI, 1, HISYM
    ds_read_u16 VT0, VT0
    TAKE CLEN
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 T0, VT0                         // insert&copy symbol
    s_cmp_lt_u32 T0, 128
    s_cselect_b32 IZ, 1, 0
    s_lshl_b32 T0, T0, 4
    s_load_dwordx4 s[92:95], IACTAB, T0                 // insert base, extra bits, copy base, extra bits
    REFILL_CHECK 1
    s_waitcnt lgkmcnt(0)
    v_bfe_u32 VE, VWINLO, 0, s93
    v_add_u32 VE, s92, VE
    TAKE s93                                            // (gfx940+: a VALU result needs 1 wait state before v_readfirstlane)
    v_readfirstlane_b32 INS, VE
    REFILL_CHECK 2
    v_bfe_u32 VE, VWINLO, 0, s95
    v_add_u32 VE, s94, VE
    TAKE s95
    v_readfirstlane_b32 CPY, VE
    REFILL_CHECK 3

// ======================================================================================================== R1
.Lr1:
    // the distance tree depends on the copy length only: request its header words now, use them after the literals
    s_sub_u32 T0, CPY, 2
    s_min_u32 T0, T0, 3                                 // distance context
    v_readlane_b32 DTREE, VDH4, T0
    s_max_i32 T0, DTREE, 0                              // (a one-symbol tree has no header: read anything)
    v_add_u32 VT0, T0, VLANE4
    ds_read_b32 VDHV, VT0
    ds_read_b32 VDHB, VT0 offset:64
    s_cmp_eq_u32 INS, 0
    s_cbranch_scc1 .Lno_lits                            // two commands in three have no literals
    s_cmp_gt_u32 INS, MBLEFT
    s_cbranch_scc1 .Lexit                               // :2036, raised by the C++ side
    s_call_b64 LINKB, .Lland
    s_sub_u32 MBLEFT, MBLEFT, INS
.Llit:
    s_sub_u32 LBLEN, LBLEN, 1
    s_cbranch_scc1 .Lx_lit_switch
    // context id -> tree descriptor, all on the vector side: A[p1] | B[p2] (pre-scaled) indexes CMH
    s_waitcnt lgkmcnt(0)
    v_or_b32 VC, VA1, VB2
    ds_read_b32 VH, VC offset:LDS_CMH
    s_waitcnt lgkmcnt(0)
    v_cmp_gt_i32 vcc, 0, VH
    s_cbranch_vccnz .Llit_single
    v_add_u32 VT0, VH, VLANE4
    ds_read_b32 VLIM, VT0
    ds_read_b32 VBASE, VT0 offset:64
    s_waitcnt lgkmcnt(0)
    LOOKUP VLIM, VBASE
    v_lshl_add_u32 VT0, VI, 1, VH
    ds_read_u16 VT2, VT0 offset:128
    TAKE CLEN
.Llit_have:
    v_mov_b32 VT0, POS
    v_add_u32 VT0, SKEW, VT0
    v_and_b32 VT0, RMASK, VT0
    v_mov_b32 VB2, VB1                                  // B[p2] of the next literal
    s_add_u32 POS, POS, 1
    s_waitcnt lgkmcnt(0)
    ds_write_b8 VT0, VT2
    ds_read_u8 VA1, VT2 offset:LDS_ATAB
    ds_read_u8 VB1, VT2 offset:LDS_BTAB
    s_cmp_ge_u32 POS, FLUSHAT
    s_cbranch_scc1 .Lflush_stub_lit
.Lflush_back_lit:
    REFILL_CHECK 4
    s_sub_u32 INS, INS, 1
    s_cmp_lg_u32 INS, 0
    s_cbranch_scc1 .Llit

.Lafter_lits:
    s_cmp_eq_u32 MBLEFT, 0
    s_cbranch_scc1 .Lexit                               // :2069 the copy part of the last command is ignored
.Lno_lits:
    s_min_u32 MAXA, POS, WINDOW
    s_cmp_lg_u32 IZ, 0
    s_cbranch_scc1 .Ldist_zero
    // ---- distance symbol (reference parse_distance_code :1367-1410)
    s_sub_u32 DBLEN, DBLEN, 1
    s_cbranch_scc1 .Lx_dist_switch
    s_cmp_lt_i32 DTREE, 0
    s_cbranch_scc1 .Ldist_single
    s_waitcnt lgkmcnt(0)
    LOOKUP VDHV, VDHB
    v_lshl_add_u32 VT0, VI, 1, DTREE
    ds_read_u16 VT2, VT0 offset:128
    TAKE CLEN
    REFILL_CHECK 5
    s_waitcnt lgkmcnt(0)
.Ldist_have:                                            // VT2 = distance code (decode_distance :1412-1481)
    v_cmp_gt_u32 vcc, 16, VT2
    s_cbranch_vccnz .Ldist_ring
    v_cmp_gt_u32 vcc, NDIR16, VT2
    s_cbranch_vccnz .Ldist_direct                       // 16 <= code < 16 + NDIRECT
    v_subrev_u32 VX, NDIR16, VT2
    v_lshrrev_b32 VN, NPOST1, VX
    v_add_u32 VN, 1, VN                                 // extra bits
    v_lshrrev_b32 VHH, NPOST, VX                        // hcode
    v_and_b32 VLC, POSTMASK, VX                         // lcode
    v_and_b32 VHH, 1, VHH
    v_add_u32 VHH, 2, VHH
    v_lshlrev_b32 VHH, VN, VHH
    v_bfe_u32 VE, VWINLO, 0, VN
    v_add3_u32 VHH, VHH, VE, -4                         // offset + extra
    v_lshlrev_b32 VHH, NPOST, VHH
    v_add3_u32 VHH, VHH, VLC, NDIRECT1
    v_lshrrev_b64 VWIN, VN, VWIN
    v_sub_u32 VNAV, VNAV, VN
    v_readfirstlane_b32 DIST, VHH
    REFILL_CHECK 6
    s_branch .Ldist_push
.Ldist_direct:
    v_readfirstlane_b32 DCODE, VT2
    s_sub_u32 DIST, DCODE, 15
    s_branch .Ldist_push
.Ldist_single:
    s_and_b32 DCODE, DTREE, 0xffff
    v_mov_b32 VT2, DCODE
    s_branch .Ldist_have
.Ldist_ring:
    v_readfirstlane_b32 DCODE, VT2
    s_cmp_eq_u32 DCODE, 0
    s_cbranch_scc1 .Ldist_zero
    s_cmp_ge_u32 DCODE, 4
    s_cbranch_scc1 .Ldist_delta
    s_mov_b32 DIST, D1
    s_cmp_eq_u32 DCODE, 2
    s_cselect_b32 DIST, D2, DIST
    s_cmp_eq_u32 DCODE, 3
    s_cselect_b32 DIST, D3, DIST
    s_branch .Ldist_push
.Ldist_delta:
    s_cmp_lt_u32 DCODE, 10
    s_cselect_b32 T0, D0, D1
    s_cselect_b32 T1, 2, 8
    s_sub_u32 T1, DCODE, T1
    s_lshr_b32 T1, T1, 1
    s_sub_u32 T2, 0, T1
    s_bitcmp1_b32 DCODE, 0
    s_cselect_b32 T1, T1, T2
    s_add_i32 DIST, T0, T1
    s_cmp_le_i32 DIST, 0
    s_cbranch_scc1 .Lx_dist_bad
.Ldist_push:
    s_cmp_gt_u32 DIST, MAXA
    s_cbranch_scc1 .Ldict                               // :1476 not pushed: static dictionary reference
    s_mov_b32 D3, D2
    s_mov_b32 D2, D1
    s_mov_b32 D1, D0
    s_mov_b32 D0, DIST
    s_branch .Lcopy
.Ldist_zero:
    s_mov_b32 DIST, D0
    s_cmp_gt_u32 DIST, MAXA
    s_cbranch_scc1 .Ldict

// ---- window copy of <= 64 bytes that does not overlap its source (copy_literals :1483-1542)
.Lcopy:
    s_min_u32 T0, DIST, 64                              // the common case: CPY <= min(64, distance, bytes left)
    s_min_u32 T0, T0, MBLEFT
    s_cmp_gt_u32 CPY, T0
    s_cbranch_scc1 .Lcopy_overlap
    s_sub_u32 T0, CPY, 1
    v_min_u32 VT0, T0, VLANE                            // switched-off lanes redo the last byte
    s_sub_u32 T1, POS, DIST
    s_cmp_gt_u32 DIST, 4096
    s_cbranch_scc1 .Lcopy_far
    s_call_b64 LINKB, .Lland_noctx                      // the source may be the pending bytes
    s_bitset0_b32 FLAGS, 2
    s_add_u32 T1, T1, SKEW
    v_mov_b32 VCLA, VT0
    v_add_u32 VT0, T1, VT0
    v_and_b32 VT0, RMASK, VT0
    ds_read_u8 VPEND, VT0
    s_branch .Lcopy_issued
.Lcopy_far:                                             // older than the ring: final in HBM, never the pending bytes
    s_sub_u32 T2, DIST, T0
    s_cmp_le_u32 T2, 4096
    s_cbranch_scc1 .Lx_r2                               // straddles the ring edge: rare
    s_bitcmp1_b32 FLAGS, 2
    s_cbranch_scc1 .Lcopy_far_b
    v_mov_b32 VCLB, VT0
    v_add_u32 VT0, T1, VT0
    buffer_load_ubyte VPENB, VT0, RSRC, 0 offen         // request first, THEN wait for and land the older copy
    s_call_b64 LINKB, .Lland_a_w1
    s_bitset1_b32 FLAGS, 2
    s_branch .Lcopy_issued
.Lcopy_far_b:
    v_mov_b32 VCLA, VT0
    v_add_u32 VT0, T1, VT0
    buffer_load_ubyte VPEND, VT0, RSRC, 0 offen
    s_call_b64 LINKB, .Lland_b_w1
    s_bitset0_b32 FLAGS, 2
.Lcopy_issued:
    s_mov_b32 PENDN, CPY
    s_add_u32 POS, POS, CPY
    s_mov_b32 PENDEND, POS                              // the pending bytes end here
    s_sub_u32 MBLEFT, MBLEFT, CPY
.Lcopy_tail:
    s_cmp_ge_u32 POS, FLUSHAT
    s_cbranch_scc1 .Lflush_stub_cmd
.Lflush_back_cmd:
    s_cmp_eq_u32 MBLEFT, 0
    s_cbranch_scc0 .Lcmd
    s_mov_b32 INS, 0
    s_branch .Lexit

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
    s_cmp_lg_u32 T3, 0
    s_cbranch_scc1 .Ldict_xform
    s_cmp_gt_u32 CPY, MBLEFT
    s_cbranch_scc1 .Lx_r2
    s_call_b64 LINKB, .Lland_noctx
    s_bitset0_b32 FLAGS, 2
    s_sub_u32 T1, CPY, 1
    v_min_u32 VCLA, T1, VLANE
    v_add_u32 VT0, T0, VCLA
    global_load_ubyte VPEND, VT0, DICTP
    s_branch .Lcopy_issued
.Llit_single:                                           // one-symbol tree: no bits
    v_and_b32 VT2, 0xff, VH
    s_branch .Llit_have

// ======================================================================================================== helpers
// Land the pending copy in the ring: its PENDN bytes sit in lanes 0..PENDN-1 of VPEND (FLAGS bit 2 clear) or VPENB (set)
// and belong just before stream position PENDEND.
// Two copies can be in flight: a far copy is requested into the free register BEFORE the older one is waited for
// (.Lland_[ab]_w1 wait with vmcnt(1): everything but the request just issued).  .Lland also refreshes the literal
// context (VA1 = A[p1], VB1 = B[p1], VB2 = B[p2]) from the last two bytes; the _noctx forms leave it stale (a copy follows, or an exit).
// (\cl = min(lane, PENDN - 1), kept from the request: lanes past the end rewrite the last byte in place, so the store
// needs no lane mask)
.macro LAND_STORE reg, cl
    s_sub_u32 T6, PENDEND, PENDN
    s_add_u32 T6, T6, SKEW
    v_add_u32 VT4, T6, \cl
    v_and_b32 VT4, RMASK, VT4
    ds_write_b8 VT4, \reg
    s_mov_b32 PENDN, 0
    s_setpc_b64 LINKB
.endm
.macro LAND_CTX reg
    s_sub_u32 T6, PENDN, 1
    v_readlane_b32 P1, \reg, T6
    s_sub_u32 T6, PENDN, 2
    v_readlane_b32 T5, \reg, T6
    v_mov_b32 VT4, P1                                   // (2 instructions after the v_readlane that wrote P1)
    ds_read_u8 VA1, VT4 offset:LDS_ATAB
    ds_read_u8 VB1, VT4 offset:LDS_BTAB
    v_mov_b32 VT4, T5
    ds_read_u8 VB2, VT4 offset:LDS_BTAB
.endm
.Lland:
    s_bitcmp1_b32 FLAGS, 1
    s_cbranch_scc1 .Lland_ringctx
    s_cmp_eq_u32 PENDN, 0
    s_cbranch_scc1 .Lland_ret
    s_waitcnt vmcnt(0) lgkmcnt(0)
    s_bitcmp1_b32 FLAGS, 2
    s_cbranch_scc1 .Lland_ctx_b
    LAND_CTX VPEND
    LAND_STORE VPEND, VCLA
.Lland_ctx_b:
    LAND_CTX VPENB
    LAND_STORE VPENB, VCLB
.Lland_noctx:
    s_cmp_eq_u32 PENDN, 0
    s_cbranch_scc1 .Lland_ret
    s_waitcnt vmcnt(0) lgkmcnt(0)
    s_bitcmp1_b32 FLAGS, 2
    s_cbranch_scc1 .Lland_store_b
    LAND_STORE VPEND, VCLA
.Lland_store_b:
    LAND_STORE VPENB, VCLB
.Lland_a_w1:
    s_cmp_eq_u32 PENDN, 0
    s_cbranch_scc1 .Lland_ret
    s_waitcnt vmcnt(1) lgkmcnt(0)
    LAND_STORE VPEND, VCLA
.Lland_b_w1:
    s_cmp_eq_u32 PENDN, 0
    s_cbranch_scc1 .Lland_ret
    s_waitcnt vmcnt(1) lgkmcnt(0)
    LAND_STORE VPENB, VCLB
.Lland_ret:
    s_setpc_b64 LINKB
// After a transformed dictionary word the last two bytes of the stream may belong to its suffix: land what is pending,
// then read them back from the ring.  (Only called from .Lr1: with fewer than 2 bytes of output, leave at R1.)
.Lland_ringctx:
    s_bitset0_b32 FLAGS, 1
    s_cmp_lt_u32 POS, 2
    s_cbranch_scc1 .Lexit
    s_mov_b64 LINKD, LINKB
    s_call_b64 LINKB, .Lland_noctx
    s_mov_b64 LINKB, LINKD
    s_add_u32 T6, POS, SKEW
    s_sub_u32 T7, T6, 1
    s_and_b32 T7, T7, RMASK
    v_mov_b32 VT0, T7
    ds_read_u8 VT0, VT0
    s_sub_u32 T7, T6, 2
    s_and_b32 T7, T7, RMASK
    v_mov_b32 VT1, T7
    ds_read_u8 VT1, VT1
    s_waitcnt lgkmcnt(0)
    ds_read_u8 VA1, VT0 offset:LDS_ATAB
    ds_read_u8 VB1, VT0 offset:LDS_BTAB
    ds_read_u8 VB2, VT1 offset:LDS_BTAB
    s_setpc_b64 LINKB

// Flush 1 KiB blocks of the ring to HBM (64 lanes x 16 B, both sides 16-byte aligned).
.Lflush:
    s_and_b32 T6, VFL, RMASK
    v_add_u32 VT4, T6, VLANE16
    ds_read_b128 VQ, VT4
    s_sub_u32 T7, VFL, SKEW
    v_add_u32 VT4, T7, VLANE16
    s_waitcnt lgkmcnt(0)
    buffer_store_dwordx4 VQ, VT4, RSRC, 0 offen
    s_add_u32 VFL, VFL, 1024
    s_add_u32 FLUSHAT, FLUSHAT, 1024
    s_cmp_ge_u32 POS, FLUSHAT
    s_cbranch_scc1 .Lflush
    s_setpc_b64 LINKC
.Lflush_stub_lit:
    s_call_b64 LINKC, .Lflush
    s_branch .Lflush_back_lit
.Lflush_stub_cmd:
    s_call_b64 LINKC, .Lflush
    s_branch .Lflush_back_cmd

// Refill reached lane WLSTOP: either the staged chunk is used up (roll the two chunks, request the next one) or
// the cursor is within 256 bits of the end of the stream (poison the block counters so that the loop leaves at
// its next R0 / R1 test; the C++ side finishes the stream with the exact end-of-input rules).
.Lspecial:
    s_cmp_lg_u32 WL, 64
    s_cbranch_scc1 .Lnear_end
    s_waitcnt vmcnt(0)
    v_mov_b32 VCHA, VCHB
    s_add_u32 CBASE, CBASE, 64
    s_add_u32 T0, CBASE, 64
    v_add_u32 VT4, T0, VLANE
    v_min_u32 VT4, WENDM1, VT4
    v_lshlrev_b32 VT4, 2, VT4
    global_load_dword VCHB, VT4, INP
    s_mov_b32 WL, 0
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

// ---- a transformed word (reference src/transformation/mod.rs, spec Appendix B): prefix + op(word) + suffix with
// op = identity / OmitFirstN / OmitLastN.  Prefix and suffix (<= 8 bytes each, from the transform record) go to the
// ring at once; the middle part is an ordinary pending load (the uppercase forms fetch it first, see .Lxf_upper).
.Ldict_xform:
    s_cmp_gt_u32 T3, 120
    s_cbranch_scc1 .Lx_r2
    s_mul_i32 T4, T3, 20                                // sizeof(BrxTransform)
    s_load_dwordx4 s[92:95], XFP, T4                    // prefix[8], suffix[8]
    s_add_u32 T4, T4, 16
    s_load_dword s97, XFP, T4                           // plen | slen << 8 | op << 16
    s_call_b64 LINKB, .Lland_noctx
    s_bitset0_b32 FLAGS, 2
    s_waitcnt lgkmcnt(0)
    s_bfe_u32 T1, s97, 0x80010                          // op
    s_and_b32 T2, s97, 0xff                             // prefix length
    s_bfe_u32 T3, s97, 0x80008                          // suffix length
    s_mov_b32 T5, 0                                     // first word byte used
    s_mov_b32 T7, CPY                                   // word bytes used
    s_bitset0_b32 FLAGS, 5                              // (bit 5: the word is already in VPEND)
    s_sub_u32 T4, T1, 1
    s_cmp_lt_u32 T4, 2
    s_cbranch_scc1 .Lxf_upper                           // UppercaseFirst / UppercaseAll
    s_cmp_lt_u32 T1, 3
    s_cbranch_scc1 .Lxf_have
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
    v_lshlrev_b32 VT1, 3, VLANE
    v_lshrrev_b64 v[20:21], VT1, s[92:93]               // lane i: prefix byte i
    s_add_u32 T4, POS, SKEW
    v_add_u32 VT0, T4, VLANE
    v_and_b32 VT0, RMASK, VT0
    s_bfm_b64 exec, T2, 0
    ds_write_b8 VT0, v20
    s_mov_b64 exec, -1
    s_add_u32 T4, T4, T2
    s_add_u32 T4, T4, T7
    v_lshrrev_b64 v[20:21], VT1, s[94:95]               // lane i: suffix byte i
    v_add_u32 VT0, T4, VLANE
    v_and_b32 VT0, RMASK, VT0
    s_bfm_b64 exec, T3, 0
    ds_write_b8 VT0, v20
    s_mov_b64 exec, -1
    s_mov_b32 PENDN, 0
    s_cmp_eq_u32 T7, 0
    s_cbranch_scc1 .Lxf_nomid
    s_mov_b32 PENDN, T7
    s_bitcmp1_b32 FLAGS, 5
    s_cbranch_scc1 .Lxf_nomid                           // uppercase forms: loaded and modified above
    s_add_u32 T0, T0, T5
    s_sub_u32 T4, T7, 1
    v_min_u32 VCLA, T4, VLANE
    v_add_u32 VT0, T0, VCLA
    global_load_ubyte VPEND, VT0, DICTP
.Lxf_nomid:
    s_add_u32 PENDEND, POS, T2
    s_add_u32 PENDEND, PENDEND, T7                      // the pending (middle) bytes end before the suffix
    s_add_u32 POS, POS, CLEN
    s_sub_u32 MBLEFT, MBLEFT, CLEN
    s_bitset1_b32 FLAGS, 1                              // literal context: from the ring (see .Lland)
    s_branch .Lcopy_tail
// UppercaseFirst / UppercaseAll (src/transformation/mod.rs:3-82): the word is fetched and waited for, one byte per lane
// (lanes past the end repeat the last byte, see LAND_STORE), and bytes are flipped as the reference does: a-z ^ 32, the
// second byte of a 2-byte UTF-8 sequence ^ 32, the third of a 3-byte one ^ 5.  A word starting with 0x00 under
// UppercaseFirst makes the reference panic (Q3): that one goes to the C++ side, which reports status 26.
.Lxf_upper:
    s_sub_u32 T4, CPY, 1
    v_min_u32 VCLA, T4, VLANE
    v_add_u32 VT0, T0, VCLA
    global_load_ubyte VPEND, VT0, DICTP
    s_bitset1_b32 FLAGS, 5
    v_mov_b32 VT1, 0                                    // xor mask per lane
    s_mov_b32 T6, 0                                     // i
    s_waitcnt vmcnt(0)
.Lxf_up_loop:
    v_readlane_b32 T4, VPEND, T6                        // b = word[i]
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
    v_cmp_eq_u32 vcc, T4, VCLA
    v_cndmask_b32 VT0, 0, VT0, vcc
    v_or_b32 VT1, VT1, VT0
    s_cmp_eq_u32 T1, 1
    s_cbranch_scc1 .Lxf_up_done                         // UppercaseFirst: one step
    s_cmp_lt_u32 T6, CPY
    s_cbranch_scc1 .Lxf_up_loop
.Lxf_up_done:
    v_xor_b32 VPEND, VPEND, VT1
    s_branch .Lxf_have

// ---- a copy of <= 64 bytes that overlaps its source (distance < length): out[i] = src[i mod distance] (:1500-1503).
// The source bytes are final (anything pending is landed first), so this is a near copy with a periodic lane index;
// lane mod distance by binary long division (lane < 64).  Longer copies go to the C++ side.
.Lcopy_overlap:
    s_min_u32 T0, MBLEFT, 64
    s_cmp_gt_u32 CPY, T0
    s_cbranch_scc1 .Lcopy_long
    s_call_b64 LINKB, .Lland_noctx
    s_bitset0_b32 FLAGS, 2
    s_sub_u32 T0, CPY, 1
    v_min_u32 VCLA, T0, VLANE
    v_mov_b32 VT0, VCLA
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
    ds_read_u8 VPEND, VT0
    s_branch .Lcopy_issued

// ---- a copy of 65..512 bytes at a distance >= 64: 64-byte chunks, each read (ring or the stream's own HBM output), waited
// for and written before the next one; nothing stays pending.  Anything longer, closer or straddling the ring edge
// goes to the C++ side (1 KiB steps, periodic fills).
.Lcopy_long:
    s_min_u32 T0, MBLEFT, 512
    s_cmp_gt_u32 CPY, T0
    s_cbranch_scc1 .Lx_r2
    s_cmp_lt_u32 DIST, 64
    s_cbranch_scc1 .Lx_r2
    s_sub_u32 T0, DIST, 4097
    s_cmp_lt_u32 T0, 63
    s_cbranch_scc1 .Lx_r2                               // 4096 < distance < 4160: a chunk would straddle the ring edge
    s_call_b64 LINKB, .Lland_noctx
    s_mov_b32 T5, CPY                                   // bytes left
.Lcl_chunk:
    s_min_u32 T0, T5, 64
    s_sub_u32 T1, T0, 1
    v_min_u32 VT1, T1, VLANE                            // clamped lane
    s_sub_u32 T1, POS, DIST
    s_cmp_gt_u32 DIST, 4096
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
    s_add_u32 POS, POS, T0
    s_sub_u32 MBLEFT, MBLEFT, T0
    s_sub_u32 T5, T5, T0
    s_cmp_ge_u32 POS, FLUSHAT
    s_cbranch_scc0 .Lcl_noflush
    s_call_b64 LINKC, .Lflush
.Lcl_noflush:
    s_cmp_lg_u32 T5, 0
    s_cbranch_scc1 .Lcl_chunk
    s_bitset1_b32 FLAGS, 1                              // literal context: from the ring (see .Lland)
    s_branch .Lflush_back_cmd

// ======================================================================================================== exits
.Lx_r0_switch:                                          // insert&copy block count exhausted (or poisoned)
    s_mov_b32 IBLEN, 0
    s_mov_b32 EXITC, 0
    s_branch .Lexit
.Lx_lit_switch:                                         // literal block count exhausted (or poisoned), mid-run
    s_mov_b32 LBLEN, 0
    s_add_u32 MBLEFT, MBLEFT, INS
    s_branch .Lexit
.Lx_dist_switch:
    s_mov_b32 DBLEN, 0
    s_mov_b32 INS, 0
    s_branch .Lexit
.Lx_dist_bad:                                           // non-positive distance: raised by the C++ side at R2
    s_mov_b32 DIST, 0
    s_mov_b32 EXITC, 3
    s_branch .Lexit
.Lx_r2:
    s_mov_b32 EXITC, 2
.Lexit:
    s_call_b64 LINKB, .Lland_noctx
    s_waitcnt vmcnt(0) lgkmcnt(0)
    // real block counters if they were poisoned
    s_bitcmp1_b32 FLAGS, 0
    s_cselect_b32 LBLEN, LBLEN_REAL, LBLEN
    s_cselect_b32 IBLEN, IBLEN_REAL, IBLEN
    // bit cursor: 32 * (CBASE + WL) - NAV
    s_add_u32 T0, CBASE, WL
    s_lshl_b32 T0, T0, 5
    v_readfirstlane_b32 T1, VNAV
    s_sub_u32 T0, T0, T1
    v_mov_b32 VT0, T0
    v_mov_b32 VT1, 0
    ds_write_b32 VZERO, VT0 offset:LDS_ST+12            // bitpos (st[3], st[4])
    ds_write_b32 VZERO, VT1 offset:LDS_ST+16
    v_mov_b32 VT0, POS
    ds_write_b32 VZERO, VT0 offset:LDS_ST+40
    v_mov_b32 VT0, VFL
    ds_write_b32 VZERO, VT0 offset:LDS_ST+48
    v_mov_b32 v20, D0
    v_mov_b32 v21, D1
    v_mov_b32 v22, D2
    v_mov_b32 v23, D3
    ds_write_b64 VZERO, v[20:21] offset:LDS_ST+56
    ds_write_b64 VZERO, v[22:23] offset:LDS_ST+64
    v_mov_b32 VT0, LBLEN
    ds_write_b32 VZERO, VT0 offset:LDS_MBW+60
    v_mov_b32 VT0, IBLEN
    ds_write_b32 VZERO, VT0 offset:LDS_MBW+84
    v_mov_b32 VT0, DBLEN
    ds_write_b32 VZERO, VT0 offset:LDS_MBW+108
    v_mov_b32 v20, MBLEFT
    v_mov_b32 v21, INS
    v_mov_b32 v22, CPY
    v_mov_b32 v23, IZ
    ds_write_b128 VZERO, v[20:23] offset:LDS_MBW+128
    s_cmp_eq_u32 EXITC, 3
    s_cselect_b32 T0, 1, 0
    s_cselect_b32 EXITC, 2, EXITC
    v_mov_b32 v20, DIST
    v_mov_b32 v21, T0
    v_mov_b32 v22, EXITC
    ds_write_b64 VZERO, v[20:21] offset:LDS_MBW+144     // distance, distance-is-bad
    ds_write_b32 VZERO, v22 offset:LDS_MBW+152          // exit point
    s_waitcnt lgkmcnt(0)
