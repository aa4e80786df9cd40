"""Hand-assembled Brotli streams (RFC 7932 bit layout) for format features no encoder at hand emits: literal context
modes LSB6 / MSB6 (libbrotlienc only uses UTF8 and SIGNED), chosen NPOSTFIX / NDIRECT with chosen distance codes.
The expected output is computed by a small independent Python model of the context rules, so these vectors pin the
oracle AND the HIP path.  Test tooling only."""
import os
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LUT = open(os.path.join(ROOT, "brotli-rs_amd", "tables", "context_lut.bin"), "rb").read()  # Lut0 | Lut1 | Lut2


class Bits:
    """LSB-first bit writer (whole bytes go to a bytearray as they fill: linear time for multi-megabyte streams)."""

    def __init__(self):
        self.buf, self.acc, self.nacc, self.n = bytearray(), 0, 0, 0

    def put(self, value, nbits):
        assert 0 <= value < (1 << nbits) or nbits == 0
        self.acc |= value << self.nacc
        self.nacc += nbits
        self.n += nbits
        if self.nacc >= 8:
            k = self.nacc >> 3
            self.buf += (self.acc & ((1 << (8 * k)) - 1)).to_bytes(k, "little")
            self.acc >>= 8 * k
            self.nacc -= 8 * k

    def put_bytes(self, data):
        assert self.nacc == 0
        self.buf += data
        self.n += 8 * len(data)

    def bytes(self):
        return bytes(self.buf) + (self.acc.to_bytes(1, "little") if self.nacc else b"")


def context_id(mode, p1, p2):
    if mode == 0:
        return p1 & 63
    if mode == 1:
        return p1 >> 2
    if mode == 2:
        return LUT[p1] | LUT[256 + p2]
    return (LUT[512 + p1] << 3) | LUT[512 + p2]


def simple_code(b, symbols, alphabet_bits):
    """simple prefix code: HSKIP = 1, NSYM - 1, the symbols (sorted, lengths 1/1, 1/2/2 or 2/2/2/2)"""
    b.put(1, 2)
    b.put(len(symbols) - 1, 2)
    for s in symbols:
        b.put(s, alphabet_bits)
    if len(symbols) == 4:
        b.put(0, 1)  # tree-select: all four codes have length 2


def code_bits(symbols, sym):
    """(bits, nbits) of `sym` in the simple code over sorted `symbols` (stream order = MSB of the code first)"""
    i = sorted(symbols).index(sym)
    n = len(symbols)
    if n == 1:
        return 0, 0
    if n == 2:
        return i, 1
    if n == 3:
        return [(0, 1), (1, 2), (3, 2)][i]  # codes 0, 10, 11 written MSB first -> LSB-first values 0, 01b, 11b
    return [(0, 2), (2, 2), (1, 2), (3, 2)][i]  # 00, 01, 10, 11 MSB-first == bit-reversed as LSB-first fields


def context_mode_stream(mode, seed, n_cmds=6):
    """One meta-block, one block type per category, NTREESL = 2 with a context map that alternates the two literal
    trees over the 64 context ids, literal trees over disjoint symbol pairs -- so every output byte reveals which
    context id the decoder computed.  Returns (stream, expected_output)."""
    rng = random.Random(seed)
    trees = [sorted(rng.sample(range(256), 2)), sorted(rng.sample(range(256), 2))]
    cmap = [rng.randrange(2) for _ in range(64)]
    # commands: insert 6 or 7 literals (insert code 6: base 6, 1 extra bit), copy 2 (code 0), distance code 0 (= last = 4)
    cmds = [(6 + rng.randrange(2),) for _ in range(n_cmds)]
    mlen = sum(c[0] for c in cmds) + 2 * (n_cmds - 1)
    b = Bits()
    b.put(0, 1)            # WBITS = 16
    b.put(1, 1)            # ISLAST
    b.put(0, 1)            # ISLASTEMPTY
    b.put(0, 2)            # MNIBBLES = 4
    b.put(mlen - 1, 16)
    b.put(0, 1); b.put(0, 1); b.put(0, 1)  # NBLTYPESL/I/D = 1
    b.put(0, 2)            # NPOSTFIX
    b.put(0, 4)            # NDIRECT
    b.put(mode, 2)         # context mode of literal block type 0
    b.put(1, 1); b.put(0, 3)  # NTREESL = 2
    b.put(0, 1)            # RLEMAX = 0
    simple_code(b, [0, 1], 1)  # context map code over 2 symbols (alphabet bits = 1)
    for c in cmap:
        b.put(c, 1)
    b.put(0, 1)            # no inverse move-to-front
    b.put(0, 1)            # NTREESD = 1
    for t in trees:
        simple_code(b, t, 8)
    iac = [176, 177]       # cell 2 (explicit distance): insert code 6, copy code 0 / 1
    simple_code(b, iac, 10)
    simple_code(b, [0], 6)  # distance code 0 only: zero-bit code
    out = bytearray()
    for k, (ins,) in enumerate(cmds):
        b.put(*code_bits(iac, 176))
        b.put(ins - 6, 1)  # insert extra bit; copy code 0 has none
        for _ in range(ins):
            p1 = out[-1] if len(out) >= 1 else 0
            p2 = out[-2] if len(out) >= 2 else 0
            t = trees[cmap[context_id(mode, p1, p2)]]
            s = rng.choice(t)
            b.put(*code_bits(t, s))
            out.append(s)
        if k != n_cmds - 1:  # copy 2 bytes from distance 4 (distance symbol has zero bits)
            for _ in range(2):
                out.append(out[-4])
    assert len(out) == mlen
    return b.bytes(), bytes(out)


# ======================================================================================================================
# General hand-assembler (round 2): complex prefix codes, dictionary references with chosen transform ids, quirk streams.
# Expected behaviour comes from the oracle (tests compare the HIP path with it); what this file guarantees is that the
# streams reach the code paths they are named after (the tests assert that through the oracle's census / status).
# ======================================================================================================================
NDBITS = [0, 0, 0, 0, 10, 10, 11, 11, 10, 10, 10, 10, 10, 9, 9, 8, 7, 7, 8, 7, 7, 6, 6, 5, 5]
INS_BASE = [0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26, 34, 50, 66, 98, 130, 194, 322, 578, 1090, 2114, 6210, 22594]
INS_EXTRA = [0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24]
CPY_BASE = [2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 18, 22, 30, 38, 54, 70, 102, 134, 198, 326, 582, 1094, 2118]
CPY_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24]
# insert&copy symbol = 64 * cell + 8 * (insert code & 7) + (copy code & 7); cell by (insert code >> 3, copy code >> 3)
CELL = {(0, 0): 2, (0, 1): 3, (1, 0): 4, (1, 1): 5, (0, 2): 6, (2, 0): 7, (1, 2): 8, (2, 1): 9, (2, 2): 10}


def rev(v, n):
    r = 0
    for i in range(n):
        r |= ((v >> i) & 1) << (n - 1 - i)
    return r


def canonical(lengths):
    """{symbol: (code, length)} of the canonical prefix code (RFC 1951 rule; symbols ascending within a length)."""
    maxlen = max(lengths) if lengths else 0
    bl = [0] * (maxlen + 2)
    for l in lengths:
        if l:
            bl[l] += 1
    code, nxt = 0, [0] * (maxlen + 2)
    for bits in range(1, maxlen + 1):
        code = (code + bl[bits - 1]) << 1
        nxt[bits] = code
    out = {}
    for s, l in enumerate(lengths):
        if l:
            out[s] = (nxt[l], l)
            nxt[l] += 1
    return out


def put_sym(b, codes, sym):
    """One symbol of a prefix code: first stream bit = MSB of the canonical code."""
    if len(codes) == 1:
        return  # one-symbol code: zero bits
    code, l = codes[sym]
    b.put(rev(code, l), l)


def _cl_code_lengths(k):
    """lengths of a complete prefix code over k >= 2 symbols, all <= 5"""
    m = max(1, (k - 1).bit_length())
    short = (1 << m) - k
    return [m - 1] * short + [m] * (k - short)


_CL_ORDER = [1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15]
_CL_FIXED = {0: (0b00, 2), 3: (0b10, 2), 4: (0b01, 2), 2: (0b011, 3), 1: (0b0111, 4), 5: (0b1111, 4)}  # LSB-first fields


def complex_code(b, lengths, zero_run_17=False, cl_len=None):
    """Complex prefix code (HSKIP = 0) for the per-symbol `lengths` (0 = absent).  Lengths after the point where the
    Kraft sum completes must be 0 and are not transmitted.  With zero_run_17 runs of >= 3 zeros use code 17.
    Returns the canonical {symbol: (code, len)} map."""
    # tokens of the code-length alphabet
    toks, total, i, n = [], 0, 0, len(lengths)
    while i < n and total < 32768:
        l = lengths[i]
        if l == 0 and zero_run_17:
            j = i
            while j < n and lengths[j] == 0:
                j += 1
            if j - i >= 3:  # chained repeat-zero codes, most significant base-8 digit first (src/lib.rs:836-872)
                reps, digits = j - i - 3, []
                while True:
                    digits.append(reps & 7)
                    reps >>= 3
                    if reps == 0:
                        break
                    reps -= 1
                for dg in reversed(digits):
                    toks.append((17, dg))
                i = j
                continue
        toks.append((l, None))
        if l:
            total += 32768 >> l
        i += 1
    assert all(l == 0 for l in lengths[i:]), "symbols after the code is complete"
    used = sorted({t[0] for t in toks})
    if cl_len is not None:  # the caller's code-length code (lengths of the 18 symbols: complete, covering every token)
        assert all(cl_len[u] for u in used) and sum(32 >> l for l in cl_len if l) == 32
        cl_len = list(cl_len)
    elif len(used) == 1:
        cl_len = [0] * 18
        cl_len[used[0]] = 1  # one code-length symbol: zero bits each (the 18 entries never sum to 32)
    else:
        cl_len = [0] * 18
        for s, l in zip(used, _cl_code_lengths(len(used))):
            cl_len[s] = l
    b.put(0, 2)  # HSKIP = 0 (kind 0 = complex, nothing skipped)
    acc = 0
    for s in _CL_ORDER:
        v, nb = _CL_FIXED[cl_len[s]]
        b.put(v, nb)
        if cl_len[s]:
            acc += 32 >> cl_len[s]
            if acc == 32:
                break
    cl_codes = canonical(cl_len)
    for s, extra in toks:
        put_sym(b, cl_codes, s)
        if s == 17:
            b.put(extra, 3)
    return canonical(list(lengths))


def uniform_lengths(alphabet, used=None):
    """A complete code over `used` (default: all) symbols of the alphabet with lengths differing by at most one."""
    used = list(range(alphabet)) if used is None else sorted(used)
    k = len(used)
    m = (k - 1).bit_length()
    short = (1 << m) - k
    lens = [0] * alphabet
    for idx, s in enumerate(used):
        lens[s] = m - 1 if idx < short else m
    # canonical order wants the short codes first by symbol value: they are (used is sorted, short ones first)
    return lens


def iac_symbol(insert_len, copy_len):
    """(symbol, insert extra (value, bits), copy extra (value, bits)) with an EXPLICIT distance (cells 2..10)."""
    ic = max(c for c in range(24) if INS_BASE[c] <= insert_len)
    cc = max(c for c in range(24) if CPY_BASE[c] <= copy_len)
    sym = 64 * CELL[(ic >> 3, cc >> 3)] + 8 * (ic & 7) + (cc & 7)
    return sym, (insert_len - INS_BASE[ic], INS_EXTRA[ic]), (copy_len - CPY_BASE[cc], CPY_EXTRA[cc])


def distance_code(distance, npostfix=0, ndirect=0):
    """(code, extra value, extra bits) of an explicit distance (no ring codes), RFC 7932 section 4."""
    if distance <= ndirect:
        return 15 + distance, 0, 0
    d = distance - ndirect - 1
    lcode = d & ((1 << npostfix) - 1)
    v = (d >> npostfix) + 4  # = offset + extra + 4 = (2 + h) << ndistbits | extra
    ndistbits = v.bit_length() - 2
    h = (v >> ndistbits) & 1
    extra = v & ((1 << ndistbits) - 1)
    hcode = 2 * (ndistbits - 1) + h
    code = 16 + ndirect + (hcode << npostfix) + lcode
    return code, extra, ndistbits


class MetaBlock:
    """One compressed meta-block with one block type per category, one literal tree (all 256 bytes, 8 bits each),
    a near-uniform insert&copy code over the symbols the commands use and a uniform distance code over the 64 symbols
    of NPOSTFIX = NDIRECT = 0.  Commands: (literal bytes, copy_len, distance) -- distance None = no copy (last command)."""

    def __init__(self, commands, mlen=None, npostfix=0, ndirect=0, lit_lengths=None, single_iac=False, single_dist=False, lit_cl_len=None):
        self.commands, self.mlen, self.npostfix, self.ndirect = commands, mlen, npostfix, ndirect
        self.single_dist = single_dist  # a ONE-symbol distance code (every copy has the same distance code; zero bits per symbol; the
                                        # copies may still differ in the code's extra bits)
        self.lit_lengths = lit_lengths  # code lengths of the 256 literals (default: 8 bits each)
        self.lit_cl_len = lit_cl_len    # ... sent with THIS code-length code (18 lengths) and every zero spelled out, instead of the default one
        self.single_iac = single_iac    # keep a ONE-symbol insert&copy code (zero bits per symbol)

    def emit(self, b, is_last, out_len_hint):
        mlen = self.mlen if self.mlen is not None else out_len_hint
        b.put(1 if is_last else 0, 1)
        if is_last:
            b.put(0, 1)  # ISLASTEMPTY
        nib = 4 if mlen <= 1 << 16 else 5 if mlen <= 1 << 20 else 6
        b.put(nib - 4, 2)
        b.put(mlen - 1, 4 * nib)
        if not is_last:
            b.put(0, 1)  # ISUNCOMPRESSED
        b.put(0, 1); b.put(0, 1); b.put(0, 1)  # NBLTYPES L / I / D = 1
        b.put(self.npostfix, 2)
        b.put(self.ndirect >> self.npostfix, 4)
        b.put(0, 2)   # context mode LSB6 (irrelevant: one tree)
        b.put(0, 1)   # NTREESL = 1
        b.put(0, 1)   # NTREESD = 1
        lit = complex_code(b, self.lit_lengths or [8] * 256, zero_run_17=self.lit_lengths is not None and self.lit_cl_len is None, cl_len=self.lit_cl_len)
        syms = sorted({iac_symbol(len(l), c if c else 2)[0] for l, c, d in self.commands})
        if len(syms) == 1 and not self.single_iac:
            syms.append(syms[0] + 1 if syms[0] + 1 < 704 else syms[0] - 1)  # (a 1-symbol insert&copy code never enters the asm loop)
        if len(syms) == 1:
            simple_code(b, syms, 10)
            iac = {syms[0]: (0, 0)}
        else:
            iac = complex_code(b, uniform_lengths(704, syms), zero_run_17=True)
        dalpha = 16 + self.ndirect + (48 << self.npostfix)
        if self.single_dist:
            codes = sorted({distance_code(d, self.npostfix, self.ndirect)[0] for l, c, d in self.commands if d is not None})
            assert len(codes) == 1
            simple_code(b, codes, max(1, (dalpha - 1).bit_length()))
            dist = {codes[0]: (0, 0)}
        else:
            dist = complex_code(b, uniform_lengths(dalpha), zero_run_17=False)
        for lits, cl, d in self.commands:
            sym, ie, ce = iac_symbol(len(lits), cl if cl else 2)
            put_sym(b, iac, sym)
            b.put(*ie)
            b.put(*ce)
            for x in lits:
                put_sym(b, lit, x)
            if d is not None:
                code, ev, eb = distance_code(d, self.npostfix, self.ndirect)
                put_sym(b, dist, code)
                b.put(ev, eb)


def raw_block(b, data):
    """Uncompressed meta-block (ISLAST = 0)."""
    assert 0 < len(data) <= 1 << 16
    b.put(0, 1)
    b.put(0, 2)
    b.put(len(data) - 1, 16)
    b.put(1, 1)  # ISUNCOMPRESSED
    b.put(0, (-b.n) % 8)
    b.put_bytes(data)


def stream_header(b, wbits=16):
    if wbits == 16:
        b.put(0, 1)
    elif wbits == 17:
        b.put(1, 7)      # 1 000 000
    elif wbits >= 18:
        b.put(1 | ((wbits - 17) << 1), 4)
    else:
        b.put(1 | ((wbits - 8) << 4), 7)  # 1 000 mmm


DOFFSET = [0] * 25
for _n in range(4, 25):
    DOFFSET[_n] = DOFFSET[_n - 1] + ((_n - 1) << NDBITS[_n - 1] if _n > 4 else 0)


def dictionary_stream(word_len, refs, seed, long_form, transform, dictionary, wbits=22, mlen_delta=0, tail=None,
                      lits_per_ref=None):
    """Dictionary references of one word length.  refs = [(transform id, word index)].  Every reference is preceded by
    1..2 literals and followed by the next command's literals, so the literal context after a transformed word is
    exercised.  long_form: a 3 KiB uncompressed meta-block first (aligns the flush cursor: the commands then run in the
    assembly loop) and 48 literal-only padding commands at the end (the assembly loop hands the last 256 bits of a
    stream to the C++ loop).  `transform(tid, word) -> bytes | None` and `dictionary` (the 122 784-byte table) are passed
    in by the test (the oracle's); the encoder needs the transformed lengths to track the output position, because a
    reference is addressed as distance = position + 1 + word id.  mlen_delta != 0 declares a wrong MLEN (Q4 tests).
    Returns (stream, expected output or None when a reference makes the reference decoder panic (Q3))."""
    rng = random.Random(seed)
    b = Bits()
    stream_header(b, wbits)
    out = bytearray()
    if long_form:
        data = bytes(rng.randrange(256) for _ in range(3072))
        raw_block(b, data)
        out += data
    start = len(out)
    nb = NDBITS[word_len]
    cmds, panics = [], False
    for tid, idx in refs:
        nl = lits_per_ref if lits_per_ref is not None else 1 + rng.randrange(2)
        lits = bytes(rng.choice(b"ab \xc3\xe4Z\x00") for _ in range(nl))
        out += lits
        idx &= (1 << nb) - 1
        word = dictionary[DOFFSET[word_len] + idx * word_len:DOFFSET[word_len] + (idx + 1) * word_len]
        cmds.append((lits, word_len, len(out) + 1 + ((tid << nb) | idx)))
        t = transform(tid, word) if tid <= 120 else b""
        if t is None:
            panics = True
            t = b""
        out += t
    ntail = (48 if long_form else 1) if tail is None else tail
    for k in range(ntail):
        lits = bytes(rng.randrange(97, 123) for _ in range(6))
        out += lits
        if k == ntail - 1:
            cmds.append((lits, 0, None))  # the meta-block ends after these literals: no copy part
        else:
            cmds.append((lits, 2, 7))     # an ordinary 2-byte copy from distance 7
            out += out[-7:-5]
    if ntail == 0:  # the last command is a reference whose word ends exactly at MLEN
        pass
    MetaBlock(cmds, mlen=len(out) - start + mlen_delta).emit(b, True, 0)
    return b.bytes(), (None if panics else bytes(out))


# ---- quirk streams: each returns a stream; the test compares the HIP path with the oracle and also pins the
# status the reading of the reference source predicts (given in the docstrings) --------------------------------------
def _mb_header(b, mlen, is_last=True):
    b.put(1 if is_last else 0, 1)
    if is_last:
        b.put(0, 1)
    b.put(0, 2)
    b.put(mlen - 1, 16)
    if not is_last:
        b.put(0, 1)


def _nbltypes(b, n):
    """parse_n_bltypes (src/lib.rs:501-525): 1 -> '0'; 2 -> '1 000'; else 1, 3-bit k, k extra bits: (1 << k) + 1 + e"""
    if n == 1:
        b.put(0, 1)
    elif n == 2:
        b.put(1, 1); b.put(0, 3)
    else:
        k = (n - 1).bit_length() - 1
        b.put(1, 1); b.put(k, 3); b.put(n - 1 - (1 << k), k)


def incomplete_code_stream(kind, pad_ones=64):
    """Q15: an under-subscribed complex prefix code is accepted when it is built; reading one of its unassigned codewords
    fails later with the error of the place that reads it.  kind -> expected status:
      'literal' 21 ParseErrorInsertLiterals, 'iac' 20 ParseErrorInsertAndCopyLength, 'distance' 19 ParseErrorDistanceCode,
      'context_map' 17 ParseErrorContextMap, 'block_type' 5 InvalidBlockSwitchCommandCode,
      'block_count' 24 UnexpectedEOF (parse_block_count maps Ok(None) to UnexpectedEOF, src/lib.rs:977).
    pad_ones: one-bits after the bad codeword (enough of them: the failure is "no such codeword", not end of input)."""
    b = Bits()
    stream_header(b, 16)
    _mb_header(b, 16)
    two_l = kind in ("block_type", "block_count")
    if two_l:
        _nbltypes(b, 2)
        if kind == "block_type":
            types = complex_code(b, [1, 2, 0, 0])          # '11' unassigned
        else:
            types = None
            simple_code(b, [0, 1], 2)                       # alphabet NBLTYPES + 2 = 4 -> 2 bits per symbol
        if kind == "block_count":
            counts = complex_code(b, [1, 2] + [0] * 24)     # '11' unassigned
            put_sym(b, counts, 0)                           # first block count: symbol 0 = 1 + 2 extra bits
        else:
            simple_code(b, [0], 5)                          # alphabet 26 -> 5 bits; one symbol: zero bits per lookup
        b.put(0, 2)                                         # extra bits of count symbol 0: block length 1
    else:
        _nbltypes(b, 1)
    _nbltypes(b, 1)
    _nbltypes(b, 1)
    b.put(0, 2); b.put(0, 4)                                # NPOSTFIX, NDIRECT
    for _ in range(2 if two_l else 1):
        b.put(0, 2)                                         # context modes
    if kind == "context_map":
        _nbltypes(b, 2)                                     # NTREESL = 2
        b.put(0, 1)                                         # RLEMAX = 0
        complex_code(b, [2, 2])                             # '1x' unassigned
        b.put(3, 2)                                         # first map entry: unassigned codeword
        b.put((1 << pad_ones) - 1, pad_ones)
        return b.bytes()
    _nbltypes(b, 1)                                         # NTREESL
    _nbltypes(b, 1)                                         # NTREESD
    lit = complex_code(b, [8] * 255 + [0] if kind == "literal" else [8] * 256)
    sym, ie, ce = iac_symbol(2, 4)
    if kind == "iac":
        iac = complex_code(b, [0] * sym + [1, 2] + [0] * (704 - sym - 2), zero_run_17=True)
    else:
        iac = complex_code(b, uniform_lengths(704, [sym, sym + 1]), zero_run_17=True)
    if kind == "distance":
        dist = complex_code(b, [1, 2] + [0] * 62, zero_run_17=True)
    else:
        dist = complex_code(b, uniform_lengths(64))
    if kind == "iac":
        b.put(3, 2)
    else:
        put_sym(b, iac, sym)
        b.put(*ie); b.put(*ce)
        if kind == "literal":
            b.put(0xFF, 8)                                  # codeword 11111111 = the absent 256th symbol
        else:
            put_sym(b, lit, 65)                             # first literal; its block count (1) is used up
            if kind == "block_type":
                b.put(3, 2)                                 # block switch before the second literal: bad type code
            elif kind == "block_count":
                put_sym(b, {0: (0, 1), 1: (1, 1)}, 1)       # type code 1 = next block type (simple code: 1 bit)
                b.put(3, 2)                                 # then the block count: unassigned codeword
            elif kind == "distance":
                put_sym(b, lit, 66)
                b.put(3, 2)
    b.put((1 << pad_ones) - 1, pad_ones)
    return b.bytes()


def metadata_skip_stream(skip_bytes_field, payload=b"after the metadata"):
    """Q2 / Q10: a metadata meta-block with MSKIPBYTES = len(skip_bytes_field) whose MSKIPLEN the reference assembles as
    byte << i (not << 8 i), then `payload` as an uncompressed meta-block and an empty last meta-block.  The metadata
    body is sized for the REFERENCE's reading, so the stream decodes to `payload` iff the decoder shares the quirk; a
    last length byte of 0 with MSKIPBYTES > 1 is UnexpectedEOF (24) in the reference (Q10)."""
    b = Bits()
    stream_header(b, 16)
    b.put(0, 1)              # ISLAST = 0
    b.put(3, 2)              # MNIBBLES code 3: metadata block
    b.put(0, 1)              # reserved
    b.put(len(skip_bytes_field), 2)
    skip = 0
    for i, x in enumerate(skip_bytes_field):
        b.put(x, 8)
        skip |= x << i       # the reference's arithmetic (src/lib.rs:460-466)
    skip += 1 if skip_bytes_field else 0
    b.put(0, (-b.n) % 8)
    for k in range(skip):
        b.put((k * 37 + 11) & 0xFF, 8)
    raw_block(b, payload)
    b.put(1, 1); b.put(1, 1)  # ISLAST, ISLASTEMPTY
    return b.bytes()


def trailer_nibble_stream(nibbles):
    """Q10: MNIBBLES = 5 or 6 with a zero top nibble -> NonZeroTrailerNibble (16), src/lib.rs:476-479."""
    b = Bits()
    stream_header(b, 16)
    b.put(1, 1); b.put(0, 1)
    b.put(nibbles - 4, 2)
    b.put(0x1234, 4 * nibbles)  # top nibble(s) zero
    b.put(0, 64)
    return b.bytes()


def farcopy_stream(seed, first=1 << 16, total=1 << 20, wbits=21):
    """The LZ77 back-reference at memory speed (BASELINE north_star "copy kernel" evidence): `first` random bytes as one
    uncompressed meta-block, then ONE compressed meta-block of back-to-back NON-overlapping copies, each doubling the
    output (copy n bytes from distance n, n = first, 2 first, ...) up to `total` bytes.  Every copied byte is read once
    from HBM and written once: physical traffic = 2 bytes per output byte (minus the first block).
    Returns (stream, expected output)."""
    rng = random.Random(seed)
    data = bytes(rng.getrandbits(8) for _ in range(first))
    b = Bits()
    stream_header(b, wbits)
    raw_block(b, data)
    out = bytearray(data)
    cmds = []
    while len(out) < total:
        n = len(out)
        cmds.append((b"", n, n))
        out += out[:n]
    MetaBlock(cmds, mlen=len(out) - first).emit(b, True, 0)
    return b.bytes(), bytes(out)


def long_stream(seed, rounds, first=1 << 16, total=1 << 20, wbits=22):
    """`rounds` x (an uncompressed meta-block of `first` fresh random bytes + a compressed meta-block of non-overlapping
    copies doubling those to `total` bytes): rounds * total bytes of output from rounds * first bytes of input, every
    back-reference within `total` bytes -- the stream of the bounded-memory Read test.  Returns (stream, expected)."""
    rng = random.Random(seed)
    b = Bits()
    stream_header(b, wbits)
    out = bytearray()
    for r in range(rounds):
        data = rng.randbytes(first)
        raw_block(b, data)
        out += data
        n, cmds = first, []
        while n < total:
            cmds.append((b"", n, n))
            n *= 2
        MetaBlock(cmds, mlen=total - first).emit(b, r == rounds - 1, 0)
        base = len(out) - first
        while len(out) - base < total:
            k = len(out) - base
            out += out[base:base + k]
    return b.bytes(), bytes(out)


def odd_long_stream(seed, rounds, first=1 << 16, mib=4, tail=2, wbits=22):
    """`rounds` x (an uncompressed meta-block of `first` random bytes + a compressed meta-block of copies: doubling to 1 MiB,
    1 MiB copies up to `mib` MiB, then ONE copy of `tail` bytes): every round is mib MiB + tail bytes, so command
    boundaries drift by `tail` bytes per round and the bounded-memory reader's window ends up over by a few bytes --
    less than its 16-byte slide granularity (the case of ADVICE r2 on brx_api.cpp bounded_step).  Returns (stream, expected)."""
    rng = random.Random(seed)
    b = Bits()
    stream_header(b, wbits)
    out = bytearray()
    total = (mib << 20) + tail
    for r in range(rounds):
        data = rng.randbytes(first)
        raw_block(b, data)
        out += data
        base = len(out) - first
        n, cmds = first, []
        while n < (1 << 20):
            cmds.append((b"", n, n))
            out += out[base:base + n]
            n *= 2
        while n < (mib << 20):
            cmds.append((b"", 1 << 20, 1 << 20))
            out += out[len(out) - (1 << 20):]
            n += 1 << 20
        cmds.append((b"", tail, 1 << 20))
        out += out[len(out) - (1 << 20):len(out) - (1 << 20) + tail]
        MetaBlock(cmds, mlen=total - first).emit(b, r == rounds - 1, 0)
    return b.bytes(), bytes(out)


def _explicit_distance(dist):
    """(distance code, number of extra bits, extra bits) of `dist` under NPOSTFIX = 0, NDIRECT = 0 (RFC 7932 section 4)"""
    for hcode in range(48):
        nbits = 1 + (hcode >> 1)
        base = ((2 + (hcode & 1)) << nbits) - 4 + 1
        if base <= dist < base + (1 << nbits):
            return 16 + hcode, nbits, dist - base
    raise ValueError(dist)


def growing_tables_stream(seed, trees_per_mb, mode=0, n_cmds=40, wbits=18, first_dist=None):
    """A stream of len(trees_per_mb) compressed meta-blocks; meta-block j has NTREESL = trees_per_mb[j] literal trees (two
    symbols each, an 18-word table + a handle per tree in the HIP decoder's table memory) behind a context map over the 64
    context ids, so the table memory a meta-block needs is chosen per meta-block: a stream whose LATER meta-block outgrows the
    kernel instance that started it (the regular one holds ~88 such trees, level 1 ~121, level 2 ~222).  Commands as in
    context_mode_stream: 6 / 7 literals, then a copy of 2 from the last distance.  first_dist: the FIRST copy of every meta-block
    but the first names that distance explicitly (and it is the last distance from then on) -- a copy from about a ring's length
    back right behind a hand-up.  Returns (stream, expected_output) -- the output from an independent model of the context rules."""
    rng = random.Random(seed)
    b = Bits()
    stream_header(b, wbits)
    out = bytearray()
    last = 4
    for j, nt in enumerate(trees_per_mb):
        is_last = j == len(trees_per_mb) - 1
        trees = [sorted(rng.sample(range(256), 2)) for _ in range(nt)]
        cmap = [rng.randrange(nt) for _ in range(64)]
        cmds = [6 + rng.randrange(2) for _ in range(n_cmds)]
        mlen = sum(cmds) + 2 * (n_cmds - 1)
        _mb_header(b, mlen, is_last)
        b.put(0, 1); b.put(0, 1); b.put(0, 1)  # NBLTYPESL/I/D = 1
        b.put(0, 2)            # NPOSTFIX
        b.put(0, 4)            # NDIRECT
        b.put(mode, 2)         # context mode of literal block type 0
        _nbltypes(b, nt)       # NTREESL (the same variable-length code as NBLTYPES)
        if nt >= 2:
            b.put(0, 1)        # RLEMAX = 0
            if nt <= 4:
                simple_code(b, list(range(nt)), max(1, (nt - 1).bit_length()))
                cm_codes = None
            else:
                cm_codes = complex_code(b, uniform_lengths(nt))
            for c in cmap:
                if cm_codes is None:
                    b.put(*code_bits(list(range(nt)), c))
                else:
                    put_sym(b, cm_codes, c)
            b.put(0, 1)        # no inverse move-to-front
        else:
            cmap = [0] * 64
        b.put(0, 1)            # NTREESD = 1
        for t in trees:
            simple_code(b, t, 8)
        iac = [176, 177]       # cell 2 (explicit distance): insert code 6, copy code 0 / 1
        simple_code(b, iac, 10)
        far = first_dist is not None and j > 0
        if far:
            dcode, dn, dx = _explicit_distance(first_dist)
            simple_code(b, [0, dcode], 6)  # one bit each: the last distance, or first_dist's code + dn extra bits
        else:
            simple_code(b, [0], 6)  # distance code 0 only: zero-bit code (= the last distance; 4 at the start of the stream)
        for k, ins in enumerate(cmds):
            b.put(*code_bits(iac, 176))
            b.put(ins - 6, 1)
            for _ in range(ins):
                p1 = out[-1] if len(out) >= 1 else 0
                p2 = out[-2] if len(out) >= 2 else 0
                t = trees[cmap[context_id(mode, p1, p2)]]
                s = rng.choice(t)
                b.put(*code_bits(t, s))
                out.append(s)
            if k != n_cmds - 1:
                if far:
                    b.put(*code_bits([0, dcode], dcode if k == 0 else 0))
                    if k == 0:
                        b.put(dx, dn)
                        last = first_dist
                for _ in range(2):
                    out.append(out[-last])
    return b.bytes(), bytes(out)


def many_trees_stream(seed, ntl, ntd, nbl_l, nbl_d, n_cmds=400, wbits=18):
    """One compressed meta-block with up to 256 literal and 256 distance trees behind context maps, nbl_l literal and nbl_d distance
    block types that SWITCH (block type symbols 0 = the one before, 1 = the next one; counts 1 .. 4 or 17 .. 24), what one piece of
    more than a megabyte looks like out of libbrotlienc (profiles/r05_big_trees.txt) at a size a test decodes in no time.  Every
    literal tree has two symbols (one bit per literal under ANY tree: the bytes that come out tell the trees apart), every distance
    tree two symbols -- one of the four last-distance codes and one explicit code of 1 or 2 extra bits; commands: 6 / 7 literals,
    then a copy of 2 .. 5 (all four distance contexts).  The caller takes the expected output from the oracle (the stream is
    valid by construction: a model of the distance ring keeps every distance inside the output)."""
    rng = random.Random(seed)
    b = Bits()
    stream_header(b, wbits)
    raw_block(b, bytes(rng.randrange(256) for _ in range(40)))  # (so that every distance of the ring and of the explicit codes is inside the output)
    cmds = [(6 + rng.randrange(2), 2 + rng.randrange(4)) for _ in range(n_cmds)]
    mlen = sum(i + c for i, c in cmds) - cmds[-1][1]  # (the last command's copy is cut off by MLEN)
    _mb_header(b, mlen, True)
    BLEN = {0: (1, 2), 4: (17, 3)}  # block count code -> (base, extra bits)

    def category(n):
        """NBLTYPES, the two codes, the first count; returns the state of the category"""
        _nbltypes(b, n)
        st = {"n": n, "cur": 0, "prev": 1, "left": 1 << 30}
        if n >= 2:
            simple_code(b, [0, 1], (n + 1).bit_length())
            simple_code(b, [0, 4], 5)
            st["left"] = count(st)
        return st

    def count(st):
        code = rng.choice((0, 0, 4))
        base, nb = BLEN[code]
        x = rng.randrange(1 << nb)
        b.put(*code_bits([0, 4], code))
        b.put(x, nb)
        return base + x

    def tick(st):
        """one symbol of the category is about to be read: the block switch in front of it, if its count is used up"""
        if st["left"] == 0:
            sym = rng.randrange(2)
            b.put(*code_bits([0, 1], sym))
            new = st["prev"] if sym == 0 else (st["cur"] + 1) % st["n"]
            st["prev"], st["cur"] = st["cur"], new
            st["left"] = count(st)
        st["left"] -= 1

    L = category(nbl_l)
    category(1)
    D = category(nbl_d)
    b.put(0, 2)  # NPOSTFIX
    b.put(0, 4)  # NDIRECT
    mode = rng.randrange(4)
    for _ in range(nbl_l):
        b.put(mode, 2)  # (one context mode: the meta-block stays in the assembly loop across literal block switches)

    def context_map(ntrees, size):
        _nbltypes(b, ntrees)
        if ntrees < 2:
            return [0] * size
        cmap = [rng.randrange(ntrees) for _ in range(size)]
        b.put(0, 1)  # RLEMAX = 0
        if ntrees <= 4:
            simple_code(b, list(range(ntrees)), max(1, (ntrees - 1).bit_length()))
            for c in cmap:
                b.put(*code_bits(list(range(ntrees)), c))
        else:
            codes = complex_code(b, uniform_lengths(ntrees))
            for c in cmap:
                put_sym(b, codes, c)
        b.put(0, 1)  # no inverse move-to-front
        return cmap

    context_map(ntl, 64 * nbl_l)
    cmap_d = context_map(ntd, 4 * nbl_d)
    for _ in range(ntl):
        simple_code(b, sorted(rng.sample(range(256), 2)), 8)
    iac = [176, 177, 178, 179]  # cell 2 (explicit distance): insert code 6, copy codes 0 .. 3 = copy lengths 2 .. 5
    simple_code(b, iac, 10)
    dtrees = []
    for _ in range(ntd):
        t = (rng.randrange(4), 16 + rng.randrange(4))  # a last-distance code, an explicit code (hcode 0 .. 3: distances 1 .. 12)
        simple_code(b, list(t), 6)
        dtrees.append(t)
    ring = [4, 11, 15, 16]  # last, second-last, ...
    pos = 40
    for k, (ins, cpy) in enumerate(cmds):
        b.put(*code_bits(iac, 176 + cpy - 2))
        b.put(ins - 6, 1)
        for _ in range(ins):
            tick(L)
            b.put(rng.randrange(2), 1)
        pos += ins
        if k == n_cmds - 1:
            break
        tick(D)
        t = dtrees[cmap_d[4 * D["cur"] + min(cpy - 2, 3)]]
        hcode = t[1] - 16
        nbits = 1 + (hcode >> 1)
        base = ((2 + (hcode & 1)) << nbits) - 4 + 1
        if ring[t[0]] <= pos and (rng.randrange(2) or base + (1 << nbits) - 1 > pos):
            b.put(*code_bits(list(t), t[0]))
            dist = ring[t[0]]
            if t[0] != 0:
                ring = [dist] + ring[:3]
        else:
            assert base + (1 << nbits) - 1 <= pos, (k, pos, t)
            x = rng.randrange(1 << nbits)
            b.put(*code_bits(list(t), t[1]))
            b.put(x, nbits)
            dist = base + x
            ring = [dist] + ring[:3]
        pos += cpy
    return b.bytes()


def periodic_stream_parts(seed, commands=700, literals=90, raw=False, single_iac=False, single_dist=False):
    """A stream of ANY length in constant memory: (prefix, unit, final, unit_output).  stream = prefix + unit * K + final decodes
    to unit_output * K.  The prefix is the stream header and an empty metadata block (which pads to a byte boundary); a unit is
    one compressed meta-block -- `commands` x (`literals` random bytes at 8 bits each, then a copy of 4 from inside the unit,
    explicit distance codes only) -- and again an empty metadata block, so every unit starts and ends on a byte boundary and
    decodes the same whatever came before it; final = ISLAST + ISLASTEMPTY.  Literal-heavy on purpose: about one compressed byte
    per output byte, so a long stream moves the reader's INPUT window as much as its output window."""
    rng = random.Random(seed)

    def empty_metadata(b):
        b.put(0, 1); b.put(3, 2); b.put(0, 1); b.put(0, 2)  # ISLAST = 0, MNIBBLES code 3, reserved, MSKIPBYTES = 0
        b.put(0, (-b.n) % 8)

    b = Bits()
    stream_header(b, 22)
    empty_metadata(b)
    prefix = b.bytes()
    if raw and literals > (1 << 20):
        # ONE uncompressed meta-block of literals - 2^20 bytes (5 length nibbles): no pause point inside it, and longer than the margin a
        # slice of the pulled reader keeps to the end of its resident input -- some of them run into that end
        data = rng.randbytes(literals - (1 << 20))
        b = Bits()
        b.put(0, 1); b.put(1, 2); b.put(len(data) - 1, 20); b.put(1, 1)  # ISLAST = 0, MNIBBLES = 5, MLEN - 1, ISUNCOMPRESSED
        b.put(0, (-b.n) % 8)
        b.put_bytes(data)
        return prefix, b.bytes(), b"\x03", data
    if raw and literals > 65536:
        # one command of `literals` bytes + a copy of 4, the literals drawn from the two 15-BIT symbols of a code with lengths
        # 1, 2, .., 14, 15, 15: almost two compressed bytes per output byte, and a single command far longer than the margin a
        # slice of the pulled reader keeps to the end of its resident input
        lens = [0] * 256
        for k in range(14):
            lens[k] = k + 1
        lens[14] = lens[15] = 15
        data = bytes(rng.choice((14, 15)) for _ in range(literals))
        b = Bits()
        MetaBlock([(data, 4, 8)], mlen=len(data) + 4, lit_lengths=lens, single_iac=single_iac, single_dist=single_dist).emit(b, False, len(data) + 4)
        empty_metadata(b)
        return prefix, b.bytes(), b"\x03", data + data[-8:-4]
    if raw:  # (raw=True: the unit is one UNCOMPRESSED meta-block of 64 KiB -- more compressed bytes than output bytes)
        data = rng.randbytes(1 << 16)
        b = Bits()
        raw_block(b, data)
        assert b.n % 8 == 0
        return prefix, b.bytes(), b"\x03", data
    cmds, out = [], bytearray()
    for k in range(commands):
        lits = rng.randbytes(literals)
        out += lits
        d = rng.randrange(4, min(len(out), 3000) + 1)
        cmds.append((lits, 4, d))
        for _ in range(4):
            out.append(out[-d])
    b = Bits()
    MetaBlock(cmds, mlen=len(out)).emit(b, False, len(out))
    empty_metadata(b)
    assert b.n % 8 == 0
    return prefix, b.bytes(), b"\x03", bytes(out)


def takeback_stream(seed, units, commands, mode=2, wbits=22, tree_syms=2):
    """Round 6: streams for the bounded reader's take-back paths.  `units` compressed meta-blocks, each with TWO literal
    trees over disjoint symbol pairs behind a context map (so every output byte tells which context the decoder computed: a literal
    decoded under the wrong tree is a wrong byte) and the same `commands`: (number of literals, copy length, distance) with explicit
    distances only (at most four different distance codes).  An insert of several KiB overwrites the decoder's whole LDS ring before
    the command's copy finds no room behind the output window (a copy of several MiB) or its literals run out of resident input (an
    insert of several hundred thousand literals: one bit each, longer than the reader's margin) -- the command is taken back and run
    again, and its first literals take their tree from the two bytes in FRONT of the command.  Returns (stream, expected output);
    the output comes from an independent model of the context rules."""
    rng = random.Random(seed)
    b = Bits()
    stream_header(b, wbits)
    out = bytearray()
    iacs = sorted({iac_symbol(n, c)[0] for n, c, d in commands})
    assert 1 < len(iacs) <= 4
    dists = sorted({_explicit_distance(d)[0] for n, c, d in commands})
    assert len(dists) <= 4
    mlen = sum(n + c for n, c, d in commands)
    nib = 4 if mlen <= 1 << 16 else 5 if mlen <= 1 << 20 else 6
    for u in range(units):
        trees = [sorted(rng.sample(range(256), tree_syms)), None]  # (4 symbols: two bits per literal -- a compression ratio of 4)
        trees[1] = sorted(rng.sample([x for x in range(256) if x not in trees[0]], tree_syms))
        cmap = [rng.randrange(2) for _ in range(64)]
        b.put(0, 1)            # ISLAST = 0
        b.put(nib - 4, 2)
        b.put(mlen - 1, 4 * nib)
        b.put(0, 1)            # ISUNCOMPRESSED = 0
        b.put(0, 1); b.put(0, 1); b.put(0, 1)  # NBLTYPESL/I/D = 1
        b.put(0, 2); b.put(0, 4)               # NPOSTFIX, NDIRECT
        b.put(mode, 2)
        _nbltypes(b, 2)        # NTREESL = 2
        b.put(0, 1)            # RLEMAX = 0
        simple_code(b, [0, 1], 1)
        for c in cmap:
            b.put(c, 1)
        b.put(0, 1)            # no inverse move-to-front
        b.put(0, 1)            # NTREESD = 1
        for t in trees:
            simple_code(b, t, 8)
        simple_code(b, iacs, 10)
        simple_code(b, dists, 6)
        for n, c, d in commands:
            sym, ie, ce = iac_symbol(n, c)
            b.put(*code_bits(iacs, sym))
            b.put(*ie)
            b.put(*ce)
            if mode == 0 and n >= 4096:
                # (fast path for the hundreds of thousands of literals of one insert: a literal costs one bit under either tree -- its
                # index in the tree -- so the bits are random bits, and the bytes follow from a 64 x 2 table of the LSB6 context)
                import numpy as np
                w = 1 if tree_syms == 2 else 2  # bits per literal; the field of sorted index i is i (two symbols) / i bit-reversed (four)
                raw = rng.randbytes((n * w + 7) // 8)
                bits = np.unpackbits(np.frombuffer(raw, dtype=np.uint8), bitorder="little")[:n * w]
                idx = (bits if w == 1 else bits[0::2] * 2 + bits[1::2]).tolist()  # (first stream bit = the code's MSB)
                tab = [tuple(trees[cmap[c]]) for c in range(64)]
                p1 = out[-1] if out else 0
                run = bytearray(n)
                for k, i in enumerate(idx):
                    p1 = tab[p1 & 63][i]
                    run[k] = p1
                out += run
                b.put(int.from_bytes(raw, "little") & ((1 << (n * w)) - 1), n * w)
            else:
                for _ in range(n):
                    p1 = out[-1] if len(out) >= 1 else 0
                    p2 = out[-2] if len(out) >= 2 else 0
                    t = trees[cmap[context_id(mode, p1, p2)]]
                    s = rng.choice(t)
                    b.put(*code_bits(t, s))
                    out.append(s)
            dcode, dn, dx = _explicit_distance(d)
            b.put(*code_bits(dists, dcode))
            b.put(dx, dn)
            assert d <= len(out)
            if d >= c:
                out += out[len(out) - d:len(out) - d + c]
            else:
                period = bytes(out[-d:])
                out += (period * (c // d + 1))[:c]
    b.put(1, 1); b.put(1, 1)   # ISLAST, ISLASTEMPTY
    return b.bytes(), bytes(out)
