// brx_small.h -- the LEAN instance of the decode kernel (included by brx_kernels.hip when BRX_SMALL is defined; brx_kernels_s.hip).
//
// 5 120 B of LDS per wave and at most 64 VGPRs: 32 single-wave workgroups per CU (the regular kernel: 16).  It is for batches of
// SHORT streams -- at most BRX_SMALL_STREAM_BYTES compressed bytes: the RLE-like fills of BASELINE configs 3 / 4 (19 and 58
// bytes in, 64 KiB and 172 KiB out), short messages -- where the regular kernel spends its time in the hand-overs between its
// out-of-line segments and in a command loop built for every meta-block shape (~12 K cycles per command,
// profiles/r03_phases.txt).  Here one function holds a stream from its first bit to its last byte:
//   * framing (reference decompress() states StreamBegin .. MetaBlockEnd, src/lib.rs:1550-1744, 2142-2167) and the
//     meta-block header (NBltypesL .. PrefixCodesDistances, :1745-2002) as straight-line code for ONE block type per category,
//     on the header path's zero-padded bit reader (hb_*): no end-of-input test per field;
//   * a command loop of its own (DataMetaBlockBegin .. CopyLiterals, :2003-2141): tables always in LDS, no block switches,
//     no resume points, the insert&copy record by one scalar load, literal context from the ring only when a run starts;
//   * long copies by the same window_copy / periodic_fill / direct_far_copy as the regular kernel.
// It decodes what is valid and plain.  EVERYTHING else -- a larger stream, more than one block type, tables beyond its 512
// words, a metadata block, any error the reference would raise, a bit read beyond the input, an output slot that is too small
// -- is LISTED for the regular kernel, which is launched right behind and decodes such a stream from its first byte with the
// reference's exact error precedence (the bytes this kernel has written are the same bytes).  So this kernel never reports
// anything but BRX_OK, and parity of every other status stays where it was.

#define SM_DEFER 0xffffu // "not for this kernel": list the stream for the regular one
FI u64 rfl64(u64 v) { return (u64)rfl((u32)v) | ((u64)rfl((u32)(v >> 32)) << 32); }

struct SmTabs { // the header's results (what Lds::mbw carries in the regular kernel)
    u32 npostfix, ndirect, cmode, ntl, ntd, cml, cmd, hl, hi, hd;
};

// Lookup in a code whose header words sit across the lanes of `hv` (loaded once per tree; layout: "Table layout in table
// memory" above) through the hb_* reader.  false = no such codeword / empty code (the reference: an error, or None).
template <bool WIDE> FI bool sm_sym(Dec &d, const Lds &s, u32 h, u32 hv, u32 &sym) {
    const u32 h0 = rdl(hv, BRX_HDR_INFO);
    const u32 kind = h0 & 3u;
    if (kind == 1u) { // one symbol: zero bits (Q5)
        sym = h0 >> 16;
        return true;
    }
    if (kind == 0u) return false;
    const u32 v = __brev(hb_peek(d) & 0x7fffu) >> 17;
    const u64 m = ballot(((v << 16) | 0xffffu) < hv) & 0xfffeull;
    if (m == 0ull) return false; // an unassigned codeword of an incomplete code (Q15)
    const u32 L = (u32)__builtin_ctzll(m);
    const u32 idx = ((v >> (15u - L)) + rdl(hv, L)) & 0xffffu;
    sym = WIDE ? rfl(s.tm[h + BRX_HDR_WORDS + idx]) : rfl(((const u16 *)s.tm)[(h + BRX_HDR_WORDS) * 2u + idx]);
    hb_skip(d, L);
    return true;
}

// Meta-block header for one block type per category.  Returns 0 or SM_DEFER.
FI u32 sm_header(Dec &d, Lds &s, SmTabs &t) {
    // NBLTYPESL / I / D (parse_n_bltypes, src/lib.rs:501-546): one bit each when all three are 1
    if (hb_bits(d, 3) != 0u) return SM_DEFER;
    t.npostfix = hb_bits(d, 2);               // :548
    t.ndirect = hb_bits(d, 4) << t.npostfix;  // :555
    t.cmode = hb_bits(d, 2);                  // :562, one literal block type
    d.lds_top = 0u;
    d.scr_top = 0u;
    const u32 lane = d.lane;
    u32 h = 0;
    // literal context map (:575, :1070-1144)
    t.ntl = read_n_bltypes(d);
    t.cml = tm_alloc(d, 16u) * 4u;
    if (lane < 16u) s.tm[(t.cml >> 2) + lane] = 0u;
    if (t.ntl >= 2u) {
        const u32 rlemax = hb_bits(d, 1) ? hb_bits(d, 4) + 1u : 0u;
        const u32 save = d.lds_top; // the map's code is dead once the map is read
        if (read_prefix_code(d, s, rlemax + t.ntl, h) || d.scr_top) return SM_DEFER;
        if (read_context_map_body(d, s, h, rlemax, t.cml, 64u)) return SM_DEFER;
        d.lds_top = save;
    }
    // distance context map (:582)
    t.ntd = read_n_bltypes(d);
    t.cmd = tm_alloc(d, 1u) * 4u;
    if (lane == 0u) s.tm[t.cmd >> 2] = 0u;
    if (t.ntd >= 2u) {
        const u32 rlemax = hb_bits(d, 1) ? hb_bits(d, 4) + 1u : 0u;
        const u32 save = d.lds_top;
        if (read_prefix_code(d, s, rlemax + t.ntd, h) || d.scr_top) return SM_DEFER;
        if (read_context_map_body(d, s, h, rlemax, t.cmd, 4u)) return SM_DEFER;
        d.lds_top = save;
    }
    // prefix codes: literals (:1016), insert&copy (:1034), distances (:1052)
    const u32 total = t.ntl + 1u + t.ntd;
    const u32 ht = tm_alloc(d, total);
    if (d.scr_top) return SM_DEFER;
    t.hl = ht; t.hi = ht + t.ntl; t.hd = ht + t.ntl + 1u;
    const u32 dalpha = 16u + t.ndirect + (48u << t.npostfix);
    for (u32 i = 0; i < total; i++) {
        const u32 alphabet = i < t.ntl ? 256u : i == t.ntl ? 704u : dalpha;
        if (read_prefix_code(d, s, alphabet, h, i > t.ntl) || d.scr_top) return SM_DEFER;
        if (lane == 0u) s.tm[ht + i] = h;
    }
    return 0u;
}

// One literal through tree `hv`; the byte goes to the ring.
#define SM_LITERAL(h_, hv_)                                                        \
    do {                                                                           \
        if (!sm_sym<false>(d, s, (h_), (hv_), lit)) return SM_DEFER;               \
        ring_put(d, s, lane == 0u, d.pos + d.a, lit);                              \
        d.pos++;                                                                   \
        if (((d.pos + d.a) & 63u) == 0u) maybe_flush(d, s);                        \
    } while (0)

// The commands of one meta-block of `mlen` bytes.  Returns 0 or SM_DEFER.
FI u32 sm_commands(Dec &d, Lds &s, const SmTabs &t, const u32 mlen, const uint4 *__restrict__ iac, const WaveConsts &wc) {
    const u32 lane = d.lane;
    const u32 h_i = rfl(s.tm[t.hi]);
    const u32 hv_i = s.tm[h_i + (lane & 31u)];
    // literal trees: one tree -> its header stays in a register; several -> context id -> tree handle as a 64 x u16 table in
    // the (now free) code-length scratch
    const u32 h_l0 = rfl(s.tm[t.hl]);
    const u32 hv_l0 = s.tm[h_l0 + (lane & 31u)];
    u16 *const cid2h = (u16 *)s.lens;
    if (t.ntl > 1u) cid2h[lane] = (u16)s.tm[t.hl + ((const u8 *)s.tm)[t.cml + lane]];
    // distance trees of the four distance contexts (min(copy_len - 2, 3), :1391); one tree -> header in a register
    const u32 cmd4 = rfl(s.tm[t.cmd >> 2]);
    const u32 h_d0 = rfl(s.tm[t.hd]);
    const u32 hv_d0 = s.tm[h_d0 + (lane & 31u)];
    u32 mb_left = mlen;
    PT_BEGIN(pc); // (bring-up build: cycles per part of a command -- slots 3 insert&copy, 4 literals, 13 distance, 14 copy)
    for (;;) {
        if (hb_over(d)) return SM_DEFER; // a read went beyond the input: UnexpectedEOF somewhere behind us
        // ---- insert&copy symbol + extra bits (parse_insert_and_copy_length :1179-1224)
        u32 sym;
        if (!sm_sym<false>(d, s, h_i, hv_i, sym)) return SM_DEFER;
        const uint4 rec = iac[sym]; // {insert base, copy base, 2 * distance context (8: implicit distance 0), extra-bit counts}
        u32 insert_len = rec.x, copy_len = rec.y;
        { const u32 n = rec.w & 0xffu; if (n) insert_len += hb_bits(d, n); }
        { const u32 n = rec.w >> 8; if (n) copy_len += hb_bits(d, n); }
        if (insert_len > mb_left) return SM_DEFER; // :2036
        PT_ADD(3, pc);
        // ---- literals (parse_insert_literals :1286-1365)
        if (insert_len) {
            u32 lit;
            if (t.ntl == 1u) {
                for (u32 k = 0; k < insert_len; k++) SM_LITERAL(h_l0, hv_l0);
            } else {
                u32 p1, p2;
                ctx_bytes(d, s, p1, p2);
                for (u32 k = 0; k < insert_len; k++) {
                    u32 cid;
                    if (t.cmode == 3u) cid = (lut8(wc.v_lut2, p1) << 3) | lut8(wc.v_lut2, p2);
                    else if (t.cmode == 2u) cid = lut8(wc.v_lut0, p1) | lut8(wc.v_lut1, p2);
                    else if (t.cmode == 0u) cid = p1 & 0x3fu;
                    else cid = p1 >> 2;
                    const u32 h = rfl((u32)cid2h[cid]);
                    const u32 hv = s.tm[h + (lane & 31u)];
                    SM_LITERAL(h, hv);
                    p2 = p1;
                    p1 = lit;
                }
            }
            mb_left -= insert_len;
            maybe_flush(d, s);
            if (mb_left == 0u) return 0u; // :2069: the copy part of the last command is ignored
        }
        PT_ADD(4, pc);
        // ---- distance (parse_distance_code :1367-1410, decode_distance :1412-1481)
        u32 distance;
        const u32 max_allowed = d.pos < d.window ? d.pos : d.window;
        if (rec.z == 8u) {
            distance = d.dist0; // implicit distance code 0 (:2012-2015): the ring stays
        } else {
            u32 dcode;
            if (t.ntd == 1u) {
                if (!sm_sym<true>(d, s, h_d0, hv_d0, dcode)) return SM_DEFER;
            } else {
                const u32 h = rfl(s.tm[t.hd + ((cmd4 >> (4u * rec.z)) & 0xffu)]); // (rec.z = 2 * context: byte `context` of the map)
                const u32 hv = s.tm[h + (lane & 31u)];
                if (!sm_sym<true>(d, s, h, hv, dcode)) return SM_DEFER;
            }
            if (dcode <= 3u) {
                distance = dcode == 0u ? d.dist0 : dcode == 1u ? d.dist1 : dcode == 2u ? d.dist2 : d.dist3;
            } else if (dcode <= 15u) {
                const u32 basev = dcode <= 9u ? d.dist0 : d.dist1;
                const u32 delta = dcode <= 9u ? (dcode - 2u) >> 1 : (dcode - 8u) >> 1;
                if ((dcode & 1u) == 0u && basev <= delta) return SM_DEFER; // non-positive
                distance = (dcode & 1u) ? basev + delta : basev - delta;
            } else if (dcode <= 15u + t.ndirect) {
                distance = dcode - 15u;
            } else {
                const u32 x = dcode - t.ndirect - 16u;
                const u32 ndistbits = 1u + (x >> (t.npostfix + 1u));
                if (ndistbits > 24u) return SM_DEFER; // (distances far beyond any window)
                const u32 e = hb_bits(d, ndistbits);
                const u32 hcode = x >> t.npostfix;
                const u32 lcode = x & ((1u << t.npostfix) - 1u);
                const u32 offset = ((2u + (hcode & 1u)) << ndistbits) - 4u;
                distance = ((offset + e) << t.npostfix) + lcode + t.ndirect + 1u;
            }
            if (dcode > 0u && distance <= max_allowed) { // :1476-1478
                d.dist3 = d.dist2; d.dist2 = d.dist1; d.dist1 = d.dist0; d.dist0 = distance;
            }
        }
        PT_ADD(13, pc);
        // ---- copy_literals :1483-1542
        if (distance <= max_allowed) {
            if (copy_len > mb_left) return SM_DEFER; // :2105
            mb_left -= copy_len;
            if (copy_len <= 64u && distance >= copy_len) {
                const u32 lc = lane < copy_len ? lane : copy_len - 1u; // switched-off lanes redo the last byte
                const u32 b = copy_fetch(d, s, distance, distance - (copy_len - 1u), distance - lc);
                ring_put(d, s, lane < copy_len, d.pos + lane + d.a, b);
                d.pos += copy_len;
                maybe_flush(d, s);
            } else {
                u32 q1, q2;
                window_copy(d, s, distance, copy_len, q1, q2);
            }
        } else {
            if (copy_len < 4u || copy_len > 24u) return SM_DEFER;
            u32 wl, wb;
            if (dict_word(d, copy_len, distance - max_allowed - 1u, wl, wb)) return SM_DEFER;
            if (wl > mb_left) return SM_DEFER;
            ring_put(d, s, lane < wl, d.pos + lane + d.a, wb);
            d.pos += wl;
            mb_left -= wl;
            maybe_flush(d, s);
        }
        PT_ADD(14, pc);
        if (mb_left == 0u) return 0u;
    }
}

// One whole stream.  Returns 0 (decoded, flushed) or SM_DEFER.
FI u32 sm_stream(Dec &d, Lds &s, const uint4 *__restrict__ iac, const WaveConsts &wc) {
    const u32 lane = d.lane;
    PT_BEGIN(ps); // (bring-up build: slots 0 staging + framing, 1 header, 2 commands, 5 final flush)
    // the whole input across the lanes of two registers, once (at most 128 dwords)
    // (the bits behind the stream's end -- the next stream's first bytes -- are masked off here, once: hb_word just reads lanes)
    {
        const u32 lastw = (u32)((d.bitend - 1ull) >> 5), r = (u32)d.bitend & 31u;
        const u32 lastmask = r ? (1u << r) - 1u : 0xffffffffu;
        d.cbase = 0u;
        d.chunkA = in_load_chunk(d, 0u) & (lane == lastw ? lastmask : 0xffffffffu);
        d.chunkB = in_load_chunk(d, 64u) & (lane + 64u == lastw ? lastmask : 0xffffffffu);
    }
    hb_begin(d);
    // parse_wbits :412-418 over the fixed tree :89-119
    u32 wbits;
    if (!hb_bits(d, 1)) {
        wbits = 16u;
    } else {
        u32 v = hb_bits(d, 3);
        if (v) {
            wbits = 17u + v;
        } else {
            v = hb_bits(d, 3);
            if (v == 1u) return SM_DEFER;
            wbits = v == 0u ? 17u : 8u + v;
        }
    }
    d.window = (1u << wbits) - 16u;
    for (;;) {
        const u32 is_last = hb_bits(d, 1);               // parse_is_last :420
        if (is_last && hb_bits(d, 1)) break;             // parse_is_last_empty :427
        const u32 nib = hb_bits(d, 2);                   // parse_m_nibbles :434
        if (nib == 3u) return SM_DEFER;                  // a metadata block
        const u32 mnibbles = nib + 4u;
        const u32 v = hb_bits(d, 16) | (mnibbles > 4u ? hb_bits(d, 4u * mnibbles - 16u) << 16 : 0u); // parse_m_len :469-483
        if (mnibbles > 4u && (v >> ((mnibbles - 1u) * 4u)) == 0u) return SM_DEFER;
        const u32 mlen = v + 1u;
        if ((u64)d.pos + mlen > (u64)d.cap) return SM_DEFER; // the slot is too small: the regular kernel says by how much
        if (!is_last && hb_bits(d, 1)) { // uncompressed :1701-1734
            u64 p = hb_pos(d);
            const u32 k = (u32)p & 7u;
            if (k && hb_bits(d, 8u - k) != 0u) return SM_DEFER;
            p = hb_pos(d);
            if (p + 8ull * mlen > d.bitend) return SM_DEFER;
            const u8 *src = (const u8 *)d.in_words + (p >> 3);
            for (u32 done = 0; done < mlen; done += 64u) {
                const u32 n = mlen - done < 64u ? mlen - done : 64u;
                const u32 b = src[done + (lane < n ? lane : n - 1u)];
                ring_put(d, s, lane < n, d.pos + lane + d.a, b);
                d.pos += n;
                maybe_flush(d, s);
            }
            d.bitpos = p + 8ull * mlen;
            hb_begin(d);
        } else {
            SmTabs t;
            PT_ADD(0, ps);
            if (sm_header(d, s, t)) return SM_DEFER;
            if (hb_over(d)) return SM_DEFER;
            PT_ADD(1, ps);
            if (sm_commands(d, s, t, mlen, iac, wc)) return SM_DEFER;
            PT_ADD(2, ps);
        }
        if (is_last) break; // MetaBlockEnd :2146-2153
    }
    // StreamEnd :2155-2167: zero bits up to the byte boundary, and nothing behind it
    {
        const u32 k = (u32)hb_pos(d) & 7u;
        if (k && hb_bits(d, 8u - k) != 0u) return SM_DEFER;
        if (hb_pos(d) != d.bitend) return SM_DEFER;
    }
    PT_ADD(0, ps);
    if (d.vfl < d.pos + d.a) flush_range(d, s, d.vfl, d.pos + d.a);
    PT_ADD(5, ps);
    return 0u;
}

// Streams are taken with a fixed stride (queue slot = workgroup index + k * grid): short streams cost about the same each,
// and 60 000 tickets from one counter would cost more than they balance ("the work queue").
__global__ __launch_bounds__(BRX_WAVE, 8) void brx_decode_kernel_s(BrxKernelArgs a) {
    Lds &s = g_lds;
    const u32 lane = threadIdx.x;
    u32 *const listed = a.work_counter + 10;
    // Classification (device-pointer path): the first ceil(n / 64) waves list the streams that are NOT for this kernel, 64
    // per atomic, in index order within a wave.
    if (a.classify != 0u) {
        for (u32 c = blockIdx.x; c * 64u < a.n; c += gridDim.x) {
            const u32 sid = c * 64u + lane;
            bool big = false;
            if (sid < a.n) {
                const u64 i0 = a.in_off[sid], i1 = a.in_off[sid + 1u];
                big = i1 < i0 || i1 - i0 > (u64)a.small_bytes;
            }
            const u64 m = ballot(big);
            if (m != 0ull) {
                const u32 base = rdl(atomicAdd(listed, lane == 0u ? (u32)__builtin_popcountll(m) : 0u), 0);
                if (big) a.s_list[base + lanes_below(m)] = sid;
                // (the sizes of the streams left to the regular kernel, summed in units of 64 B: it starts the longer half of an
                // oversubscribed queue first -- brx_kernels.hip, two_walk)
                u32 units = 0u;
                if (big && sid < a.n) { const u64 len = a.in_off[sid + 1u] - a.in_off[sid]; units = len >> 6 > 0xfffffull ? 0xfffffu : (u32)(len >> 6); }
                u32 umax = units, umin = big && sid < a.n ? 0xfffffu - units : 0u; // (0xfffff - size: the smallest as a maximum)
                for (u32 o = 32u; o != 0u; o >>= 1) {
                    units += (u32)__shfl_xor((int)units, (int)o);
                    umax = max(umax, (u32)__shfl_xor((int)umax, (int)o));
                    umin = max(umin, (u32)__shfl_xor((int)umin, (int)o));
                }
                if (lane == 0u) {
                    (void)atomicAdd(a.work_counter + 15, units);
                    (void)atomicMax(a.work_counter + 16, umax);
                    (void)atomicMax(a.work_counter + 17, umin);
                }
            }
        }
    }
    const WaveConsts wc = wave_consts((const u32 *)a.t.context_lut);
    const uint4 *__restrict__ iac = (const uint4 *)a.t.iac;
    for (u32 slot = blockIdx.x; slot < a.n; slot += gridDim.x) {
        // (everything that becomes decoder state goes through readfirstlane: values LLVM cannot prove wave-uniform would put the
        // buffer resources into VGPRs and a waterfall loop around every store)
        const u32 sid = rfl(a.order != nullptr ? a.order[slot] : slot);
        const u64 i0 = rfl64(a.in_off[sid]), i1 = rfl64(a.in_off[sid + 1u]);
        const u64 o0 = rfl64(a.out_off[sid]), o1 = rfl64(a.out_off[sid + 1u]);
        const bool big = i1 < i0 || i1 - i0 > (u64)a.small_bytes;
        u32 rc = SM_DEFER;
        if (!big) {
            Dec d;
            d.lane = lane;
            const u8 *inp = a.in + i0;
            const u32 mis = (u32)((uintptr_t)inp & 3u);
            d.in_words = (const u32 *)(inp - mis);
            const u32 in_len = (u32)(i1 - i0);
            d.w_end = (mis + in_len + 3u) >> 2;
            d.bitend = 8ull * (mis + in_len);
            d.bitpos = 8ull * mis;
            d.out = a.out + o0;
            d.mirror = a.out_mirror != nullptr ? a.out_mirror + o0 : nullptr;
            const u64 capacity = o1 >= o0 ? o1 - o0 : 0ull;
            d.cap = capacity > 0xffffff00ull ? 0xffffff00u : (u32)capacity;
            d.out_rsrc = __builtin_amdgcn_make_buffer_rsrc(d.out, 0, d.cap, 0x00020000);
            d.out2_rsrc = __builtin_amdgcn_make_buffer_rsrc(d.mirror, 0, d.mirror ? d.cap : 0u, 0x00020000);
            d.pos = 0;
            d.a = (u32)((uintptr_t)d.out & 15u);
            d.vfl = d.a;
            d.window = 0;
            d.dist0 = 4; d.dist1 = 11; d.dist2 = 15; d.dist3 = 16; // src/lib.rs:408
            d.lds_top = 0;
            d.scr_top = 0;
            d.scratch = nullptr;
            d.pool = nullptr;
            d.t_dict = a.t.dict;
            d.t_xforms = a.t.xforms;
            d.t_lut = (const u32 *)a.t.context_lut;
#ifdef BRX_BRINGUP
            if (lane < 32u) g_prof[lane] = 0ull;
            const unsigned long long t_stream = __builtin_readcyclecounter();
#endif
            rc = in_len != 0u ? sm_stream(d, s, iac, wc) : SM_DEFER;
#ifdef BRX_BRINGUP
            if (a.debug != nullptr && rc == 0u) {
                if (lane < 32u) a.debug[(size_t)a.n_total * 10u + (size_t)sid * 32u + lane] = g_prof[lane];
                if (lane == 0u) a.debug[(size_t)sid * 10u + 8] = __builtin_readcyclecounter() - t_stream;
            }
#endif
            if (rc == 0u && lane == 0u) {
                a.status[sid] = (int)ST_OK;
                a.out_len[sid] = (u64)d.pos;
            }
        }
        // not decoded here: a small stream is listed now; a large one was listed by the classification (or, without one, was
        // never in this kernel's queue -- should it be, it is listed like a small one)
        if (rc != 0u && (!big || a.classify == 0u)) {
            const u32 at = rdl(atomicAdd(listed, lane == 0u ? 1u : 0u), 0);
            if (lane == 0u) a.s_list[at] = sid;
        }
    }
}

void brx_launch_decode_s(const BrxKernelArgs &args, unsigned grid, void *hip_stream) {
    hipLaunchKernelGGL(brx_decode_kernel_s, dim3(grid), dim3(BRX_WAVE), 0, (hipStream_t)hip_stream, args);
}
