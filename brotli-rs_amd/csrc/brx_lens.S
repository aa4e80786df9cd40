// brx_lens.S -- the code-length symbol loop of a complex prefix code (reference parse_complex_prefix_code, the symbol part:
// src/lib.rs:739-874), hand-written for gfx950.  Preprocessed by build.py (register names) and pasted into ONE asm statement of
// read_complex_lens() in brx_kernels.hip; `@n@` stands for operand n of that statement.
//
// The compiled C++ form of this loop took 300 cycles per symbol (LLVM turns its early exits into chains of lane-mask
// booleans); alice29 sends 1 500 of them per meta-block.  Here: ~22 instructions per plain length.
//
// The bit reader is the header path's (hb_* in brx_kernels.hip): a 64-bit window with >= 32 valid bits over the input padded
// with zeros behind its last real bit, refilled a dword at a time from the two staged chunks (lane k of VCHA / VCHB = dword
// CBASE + k / CBASE + 64 + k); NO end-of-input test here (one check behind the header).  The 64 lengths of the chunk in
// progress collect in the lanes of VCUR and go to Lds::lens with one byte store per lane when the chunk is left.
//
// operands: 0 status (out, SGPR)          1 nz (out)                2 i (out)                 3 dirty (out)
//           4 window (in/out, SGPR pair)  5 nav (in/out)            6 ww (in/out)             7 cbase (in/out)
//           8 VCHA (in/out)  9 VCHB (in/out)  10 VCUR (out)  11, 12 VGPR temporaries
//           13 alphabet  14 w_end  15 last dword  16 its mask  17 in_words (SGPR pair)  18 cltab (VGPR)  19 lane id (VGPR)
// status: 0 = the lengths are complete (alphabet full, or the code space used up), else the reference's error
// (1 = CodeLengthsChecksum, 18 = ParseErrorComplexPrefixCodeLengths).  The caller stores the last chunk (VCUR, dirty) itself.
#define WIN s[60:61]
#define WINLO s60
#define NAV s62
#define WW s63
#define IPOS s64
#define TOT s65
#define NZ s66
#define LNZ s67
#define LSYM s68
#define LREP s69
#define ALPHA s70
#define CBASE s71
#define WEND s72
#define LASTW s73
#define LMASK s74
#define ENT s75
#define SYM s76
#define LEN s77
#define T0 s78
#define T1 s79
#define PAIR s[80:81]
#define PLO s80
#define PHI s81
#define T2 s82
#define T3 s83
#define LINK s[84:85]
#define INW s[86:87]
#define STAT s88
#define DIRTY s89
#define LEFT s90
#define VCHA @8@
#define VCHB @9@
#define VCUR @10@
#define VT0 @11@
#define VT1 @12@
#define VCLTAB @18@
#define VLANE @19@

    s_mov_b64 WIN, @4@
    s_mov_b32 NAV, @5@
    s_mov_b32 WW, @6@
    s_mov_b32 CBASE, @7@
    s_mov_b32 ALPHA, @13@
    s_mov_b32 WEND, @14@
    s_mov_b32 LASTW, @15@
    s_mov_b32 LMASK, @16@
    s_mov_b64 INW, @17@
    s_mov_b32 IPOS, 0
    s_mov_b32 TOT, 0
    s_mov_b32 NZ, 0
    s_mov_b32 LNZ, 8
    s_mov_b32 LSYM, 0xff
    s_mov_b32 LREP, 0
    s_mov_b32 DIRTY, 0
    s_mov_b32 STAT, 0
    v_mov_b32 VCUR, 0
.Lls_sym:
    s_cmp_ge_u32 IPOS, ALPHA
    s_cbranch_scc1 .Lls_done
    s_and_b32 T0, WINLO, 31
    v_readlane_b32 ENT, VCLTAB, T0                      // (symbol << 4) | code length
    s_and_b32 LEN, ENT, 15
    s_lshr_b32 SYM, ENT, 4
    s_lshr_b64 WIN, WIN, LEN
    s_sub_u32 NAV, NAV, LEN
    s_cmp_lt_u32 NAV, 32
    s_cbranch_scc0 .Lls_have
    s_call_b64 LINK, .Lls_refill
.Lls_have:
    s_cmp_gt_u32 SYM, 15
    s_cbranch_scc1 .Lls_special
    // ---- a plain length 0..15
    s_and_b32 T0, IPOS, 63
    v_mov_b32 VT0, SYM
    v_cmp_eq_u32 vcc, T0, VLANE
    v_cndmask_b32 VCUR, VCUR, VT0, vcc
    s_add_u32 IPOS, IPOS, 1
    s_mov_b32 LSYM, SYM
    s_mov_b32 LREP, 0
    s_cmp_eq_u32 SYM, 0
    s_cbranch_scc1 .Lls_chunk
    s_mov_b32 DIRTY, 1
    s_add_u32 NZ, NZ, 1
    s_mov_b32 LNZ, SYM
    s_lshr_b32 T1, 0x8000, SYM
    s_add_u32 TOT, TOT, T1
    s_cmp_ge_u32 TOT, 0x8000
    s_cbranch_scc1 .Lls_total
.Lls_chunk:
    s_and_b32 T0, IPOS, 63
    s_cmp_lg_u32 T0, 0
    s_cbranch_scc1 .Lls_sym
    s_sub_u32 T0, IPOS, 64
    s_call_b64 LINK, .Lls_flush
    s_branch .Lls_sym
.Lls_total:                                             // the code space is used up exactly (done) or overrun (error)
    s_cmp_eq_u32 TOT, 0x8000
    s_cbranch_scc1 .Lls_done
    s_mov_b32 STAT, 1
    s_branch .Lls_done
.Lls_special:
    s_cmp_eq_u32 SYM, 16
    s_cbranch_scc1 .Lls_rep
    // ---- 17: a run of zeros, 3 extra bits; a 17 right behind a 17 extends the run (src/lib.rs:839-860)
    s_and_b32 T0, WINLO, 7
    s_lshr_b64 WIN, WIN, 3
    s_sub_u32 NAV, NAV, 3
    s_cmp_lt_u32 NAV, 32
    s_cbranch_scc0 .Lls_z_have
    s_call_b64 LINK, .Lls_refill                        // (T0 survives: the refill uses T1, T2, PAIR)
.Lls_z_have:
    s_cmp_eq_u32 LSYM, 17
    s_cselect_b32 T1, LREP, 0
    s_cmp_eq_u32 T1, 0
    s_cbranch_scc1 .Lls_z_fresh
    s_sub_u32 T2, T1, 2
    s_lshl_b32 T2, T2, 3
    s_add_u32 T2, T2, T0
    s_add_u32 T2, T2, 3                                 // new repeat = 8 * (old - 2) + extra + 3
    s_sub_u32 T3, T2, T1                                // zeros added by this symbol
    s_mov_b32 LREP, T2
    s_branch .Lls_z_go
.Lls_z_fresh:
    s_add_u32 T3, T0, 3
    s_mov_b32 LREP, T3
.Lls_z_go:
    s_mov_b32 LSYM, 17
    s_add_u32 T2, IPOS, T3
    s_cmp_gt_u32 T2, ALPHA
    s_cbranch_scc1 .Lls_err_parse
    s_xor_b32 T0, T2, IPOS
    s_cmp_lt_u32 T0, 64
    s_cbranch_scc1 .Lls_z_same                          // (same chunk)
    s_and_b32 T0, IPOS, 0xffffffc0
    s_mov_b32 LEFT, T2
    s_call_b64 LINK, .Lls_flush                         // leave the chunk: the ones in between stay zero (lens_clear)
    s_mov_b32 T2, LEFT
.Lls_z_same:
    s_mov_b32 IPOS, T2
    s_branch .Lls_sym
.Lls_rep:
    // ---- 16: the last non-zero length again, 2 extra bits; a 16 right behind a 16 extends the run (:812-838)
    s_and_b32 T0, WINLO, 3
    s_lshr_b64 WIN, WIN, 2
    s_sub_u32 NAV, NAV, 2
    s_cmp_lt_u32 NAV, 32
    s_cbranch_scc0 .Lls_r_have
    s_call_b64 LINK, .Lls_refill
.Lls_r_have:
    s_cmp_eq_u32 LSYM, 16
    s_cselect_b32 T1, LREP, 0
    s_cmp_eq_u32 T1, 0
    s_cbranch_scc1 .Lls_r_fresh
    s_sub_u32 T2, T1, 2
    s_lshl_b32 T2, T2, 2
    s_add_u32 T2, T2, T0
    s_add_u32 T2, T2, 3                                 // new repeat = 4 * (old - 2) + extra + 3
    s_sub_u32 LEFT, T2, T1
    s_branch .Lls_r_go
.Lls_r_fresh:
    s_add_u32 T2, T0, 3
    s_mov_b32 LEFT, T2
.Lls_r_go:                                              // T2 = new repeat, LEFT = lengths added by this symbol
    s_add_u32 T3, IPOS, LEFT
    s_cmp_gt_u32 T3, ALPHA
    s_cbranch_scc1 .Lls_err_parse
    s_add_u32 NZ, NZ, LEFT
    s_lshr_b32 T3, 0x8000, LNZ
    s_mul_i32 T3, T3, LEFT
    s_add_u32 TOT, TOT, T3
    s_mov_b32 LREP, T2                                  // (the reference sets these behind the checks below: nobody reads them
    s_mov_b32 LSYM, 16                                  // when a check ends the loop)
    v_mov_b32 VT1, LNZ
.Lls_r_fill:                                            // the run, chunk by chunk
    s_and_b32 T0, IPOS, 63
    s_sub_u32 T1, 64, T0
    s_min_u32 T1, T1, LEFT                              // lengths of the run inside this chunk
    v_subrev_u32 VT0, T0, VLANE                         // lane - first lane of the run
    v_cmp_gt_u32 vcc, T1, VT0
    v_cndmask_b32 VCUR, VCUR, VT1, vcc
    s_mov_b32 DIRTY, 1
    s_add_u32 IPOS, IPOS, T1
    s_sub_u32 LEFT, LEFT, T1
    s_and_b32 T0, IPOS, 63
    s_cmp_lg_u32 T0, 0
    s_cbranch_scc1 .Lls_r_next
    s_sub_u32 T0, IPOS, 64
    s_call_b64 LINK, .Lls_flush
    v_mov_b32 VT1, LNZ
.Lls_r_next:
    s_cmp_lg_u32 LEFT, 0
    s_cbranch_scc1 .Lls_r_fill
    s_cmp_ge_u32 TOT, 0x8000
    s_cbranch_scc1 .Lls_total
    s_branch .Lls_sym
.Lls_err_parse:
    s_mov_b32 STAT, 18
    s_branch .Lls_done

// ---- store the chunk at symbol T0 (if anything non-zero was put into it) and start an empty one.  Clobbers VT0.
.Lls_flush:
    s_cmp_eq_u32 DIRTY, 0
    s_cbranch_scc1 .Lls_flush_empty
    v_add_u32 VT0, T0, VLANE
    ds_write_b8 VT0, VCUR offset:LDS_LENS
.Lls_flush_empty:
    v_mov_b32 VCUR, 0
    s_mov_b32 DIRTY, 0
    s_setpc_b64 LINK

// ---- the next dword of the input enters the window (NAV < 32 valid bits).  Clobbers T1, T2, PAIR, VT0, VT1, vcc.
.Lls_refill:
    s_sub_u32 T1, WW, CBASE
    s_cmp_lt_u32 T1, 128
    s_cbranch_scc1 .Lls_rf_staged
    // roll the staged chunks: B becomes A, the next 64 dwords (zeros from the stream's last dword on) become B
    s_waitcnt vmcnt(0)
    v_mov_b32 VCHA, VCHB
    s_add_u32 CBASE, CBASE, 64
    s_add_u32 T1, CBASE, 64
    v_add_u32 VT0, T1, VLANE
    v_cmp_gt_u32 vcc, WEND, VT0
    s_sub_u32 T2, WEND, 1
    v_min_u32 VT1, T2, VT0
    v_lshlrev_b32 VT1, 2, VT1
    global_load_dword VCHB, VT1, INW
    s_waitcnt vmcnt(0)
    v_cndmask_b32 VCHB, 0, VCHB, vcc
    s_sub_u32 T1, WW, CBASE
.Lls_rf_staged:
    s_cmp_lt_u32 T1, 64
    s_cbranch_scc0 .Lls_rf_b
    v_readlane_b32 PLO, VCHA, T1
    s_branch .Lls_rf_have
.Lls_rf_b:
    v_readlane_b32 PLO, VCHB, T1                        // (the lane select takes the low six bits)
.Lls_rf_have:
    s_cmp_lt_u32 WW, WEND
    s_cselect_b32 PLO, PLO, 0
    s_cmp_eq_u32 WW, LASTW
    s_cselect_b32 T2, LMASK, -1
    s_and_b32 PLO, PLO, T2
    s_mov_b32 PHI, 0
    s_lshl_b64 PAIR, PAIR, NAV
    s_or_b64 WIN, WIN, PAIR
    s_add_u32 NAV, NAV, 32
    s_add_u32 WW, WW, 1
    s_setpc_b64 LINK

.Lls_done:
    s_mov_b32 @0@, STAT
    s_mov_b32 @1@, NZ
    s_mov_b32 @2@, IPOS
    s_mov_b32 @3@, DIRTY
    s_mov_b64 @4@, WIN
    s_mov_b32 @5@, NAV
    s_mov_b32 @6@, WW
    s_mov_b32 @7@, CBASE
