"""Hand-assembled Brotli streams (RFC 7932 bit layout) for format features no encoder at hand emits: literal context
modes LSB6 / MSB6 (libbrotlienc only uses UTF8 and SIGNED), chosen NPOSTFIX / NDIRECT with chosen distance codes.
The expected output is computed by a small independent Python model of the context rules, so these vectors pin the
oracle AND the HIP path.  Test tooling only."""
import os
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LUT = open(os.path.join(ROOT, "brotli-rs_amd", "tables", "context_lut.bin"), "rb").read()  # Lut0 | Lut1 | Lut2


class Bits:
    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, value, nbits):
        assert 0 <= value < (1 << nbits) or nbits == 0
        self.v |= value << self.n
        self.n += nbits

    def bytes(self):
        return self.v.to_bytes((self.n + 7) // 8, "little")


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
