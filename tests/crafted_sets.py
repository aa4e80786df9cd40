"""Named sets of hand-assembled streams (tests/craft.py) shared by the CPU tests (oracle pins the predicted behaviour)
and the GPU tests (HIP path vs oracle).  Each entry: (name, stream, expected_status, expected_output or None)."""
import random

import craft
import oracle_py

UPPERCASE_FIRST = [4, 9, 15, 30, 58, 65, 66, 69, 74, 78, 79, 88, 91, 96, 99, 104, 108, 109, 118, 120]
OMIT_FIRST = [3, 11, 26, 34, 39, 40, 54, 55]


def _dictionary():
    return bytes(oracle_py.lib().bro_dictionary()[:122784])


def zero_led_words(D):
    """(word length, index) of the dictionary words that start with 0x00 (Q3)."""
    out = []
    for L in range(4, 25):
        for i in range(1 << craft.NDBITS[L]):
            if D[craft.DOFFSET[L] + i * L] == 0:
                out.append((L, i))
    return out


def transform_streams():
    """Every transform id 0..120 on every word length 4..24, short form (C++ loop) and long form (assembly loop), words
    without a leading 0x00.  OmitFirstN with N >= len - 1 (Q1) happens at lengths 4..9, OmitLastN emptying the word
    (zero-length result) at lengths <= 9."""
    D = _dictionary()
    bad = set(zero_led_words(D))
    sets = []
    for L in range(4, 25):
        rng = random.Random(1000 + L)
        refs = []
        for tid in range(121):
            while True:
                idx = rng.randrange(1 << craft.NDBITS[L])
                if (L, idx) not in bad:
                    break
            refs.append((tid, idx))
        rng.shuffle(refs)
        for long_form in (False, True):
            s, e = craft.dictionary_stream(L, refs, 7 * L + long_form, long_form, oracle_py.transform, D)
            assert e is not None
            sets.append(("xf_len%d_%s" % (L, "long" if long_form else "short"), s, 0, e))
    return sets


def transform_edge_streams():
    """Q3 (UppercaseFirst on a 0x00-led word -> status 26), Q4 (transformed length against MLEN: ends exactly at MLEN ->
    ok, one byte short -> ExceededExpectedBytes 3), transform id > 120 -> 9, 0x00-led words under the other transforms
    (UppercaseAll handles 0x00), words with no literal between two references."""
    D = _dictionary()
    zl = zero_led_words(D)
    sets = []
    for k, (L, idx) in enumerate(zl):
        for long_form in (False, True):
            tid = UPPERCASE_FIRST[k % len(UPPERCASE_FIRST)]
            refs = [(0, 3), (44, 5), (tid, idx), (1, 9)]
            s, e = craft.dictionary_stream(L, refs, 50 + k, long_form, oracle_py.transform, D)
            assert e is None
            sets.append(("q3_len%d_idx%d_%s" % (L, idx, "long" if long_form else "short"), s, 26, None))
        refs = [(t, idx) for t in range(121) if t not in UPPERCASE_FIRST]
        s, e = craft.dictionary_stream(L, refs, 90 + k, True, oracle_py.transform, D)
        sets.append(("zero_led_other_transforms_%d_%d" % (L, idx), s, 0, e))
    for L in (4, 7, 12, 24):
        for long_form in (False, True):
            refs = [(t, 17 * t + 1) for t in (0, 12, 23, 3, 49, 64, 73, 120, 44)]
            tag = "len%d_%s" % (L, "long" if long_form else "short")
            s, e = craft.dictionary_stream(L, refs, 300 + L, long_form, oracle_py.transform, D, tail=0)
            sets.append(("ends_at_mlen_" + tag, s, 0, e))
            s, e = craft.dictionary_stream(L, refs, 300 + L, long_form, oracle_py.transform, D, tail=0, mlen_delta=-1)
            sets.append(("exceeds_mlen_" + tag, s, 3, None))
            s, e = craft.dictionary_stream(L, [(0, 1), (121, 2), (5, 3)], 400 + L, long_form, oracle_py.transform, D)
            sets.append(("transform_id_121_" + tag, s, 9, None))
            s, e = craft.dictionary_stream(L, refs, 500 + L, long_form, oracle_py.transform, D, lits_per_ref=0)
            sets.append(("back_to_back_" + tag, s, 0, e))
    return sets


def quirk_streams():
    sets = []
    for kind, st in (("literal", 21), ("iac", 20), ("distance", 19), ("context_map", 17), ("block_type", 5),
                     ("block_count", 24)):
        sets.append(("q15_" + kind, craft.incomplete_code_stream(kind), st, None))
        sets.append(("q15_" + kind + "_eof", craft.incomplete_code_stream(kind, pad_ones=0), None, None))
        sets.append(("q15_" + kind + "_short_pad", craft.incomplete_code_stream(kind, pad_ones=3), None, None))
    payload = b"after the metadata"
    for f in (b"", b"\x05", b"\x03\x01", b"\x02\x01\x01", b"\xff\xff", b"\x10\x20\x30", b"\x80\x01", b"\x00\x00\x01"):
        sets.append(("q2_mskip_" + f.hex(), craft.metadata_skip_stream(f, payload), 0, payload))
    for f in (b"\x07\x00", b"\x01\x02\x00"):
        sets.append(("q10_mskip_last_zero_" + f.hex(), craft.metadata_skip_stream(f, payload), 24, None))
    for n in (5, 6):
        sets.append(("q10_trailer_nibble_%d" % n, craft.trailer_nibble_stream(n), 16, None))
    return sets


def boundary_streams():
    """Copies at the two limits the assembly loop checks lazily: a copy that runs 1..3 bytes past MLEN (taken back after
    it was issued, raised by the C++ side) -- with and without literals in front, i.e. with the copy's lanes pending or
    landed -- and distances equal to the bytes produced so far (the largest window reference) and one more (the first
    dictionary word, or past the window) after a long stretch of short distances (the loop's cached distance bound is
    stale then).  Expected status / bytes come from the oracle."""
    import random
    from craft import Bits, MetaBlock, raw_block, stream_header
    sets = []
    for seed, over in enumerate((1, 2, 3, 7)):
        for lits_first in (0, 2):
            rng = random.Random(900 + seed)
            head = rng.randbytes(300)
            cmds, n = [], len(head)
            for k in range(400):
                lit = rng.randbytes(lits_first if k % 3 else 0)
                cl = rng.choice((2, 3, 4, 5, 7, 9))
                cmds.append((lit, cl, rng.randrange(1, min(n, 1500) + 1) if k % 5 else rng.randrange(cl, n + 1)))
                n += len(lit) + cl
            b = Bits()
            stream_header(b, 18)
            raw_block(b, head)
            MetaBlock(cmds, mlen=n - len(head) - over).emit(b, True, 0)
            # (64 more bytes behind: the assembly loop hands the last 256 bits of a stream to the C++ side, and this is about the loop)
            sets.append(("copy_past_mlen_%d_lits%d" % (over, lits_first), b.bytes() + rng.randbytes(64), None, None))
            sets.append(("copy_past_mlen_%d_lits%d_at_eof" % (over, lits_first), b.bytes(), None, None))
    for delta in (0, 1, 2):
        for cl in (4, 6):
            rng = random.Random(950 + delta)
            head = rng.randbytes(2000)
            cmds, n = [], len(head)
            for k in range(300):
                if k % 50 == 49:
                    cmds.append((b"xy", cl, n + 2 + delta))   # distance = position (+ delta) at the copy
                else:
                    cmds.append((rng.randbytes(k % 3), 3, rng.randrange(3, 64)))
                n += len(cmds[-1][0]) + cmds[-1][1]
            b = Bits()
            stream_header(b, 18)
            raw_block(b, head)
            MetaBlock(cmds, mlen=n - len(head)).emit(b, True, 0)
            sets.append(("distance_at_position_plus_%d_len%d" % (delta, cl), b.bytes(), None, None))
    return sets


def growing_table_streams():
    """Streams whose meta-blocks differ in the table memory they need (craft.growing_tables_stream): later meta-blocks outgrow the
    kernel instance that started the stream, or the first one does, or none.  Expected bytes from craft.py's own model."""
    import craft
    sets = []
    for i, shape in enumerate(([2, 2, 105, 2, 150], [1, 3, 5, 250, 2], [100, 2], [2] * 6, [100, 2, 160], [3, 140], [170, 1, 1, 110], [256, 2])):
        for mode in (i % 4, (i + 1) % 4):
            st, out = craft.growing_tables_stream(60 + i, shape, mode=mode, n_cmds=25 + 30 * (i % 3))
            sets.append(("growing_tables_%s_mode%d" % ("_".join(map(str, shape)), mode), st, 0, out))
    return sets


def all_sets():
    return transform_streams() + transform_edge_streams() + quirk_streams() + boundary_streams() + growing_table_streams()
