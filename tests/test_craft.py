"""CPU tests of the hand-assembled streams (tests/craft.py, tests/crafted_sets.py): the oracle, in both prefix-lookup
modes, must give the status (and bytes) each stream was built to produce.  This pins the crafting tool and the oracle's
reading of the unpinned quirks Q1/Q2/Q3/Q10/Q15 against each other; the GPU tests then hold the HIP path to the same."""
import os
import subprocess
import sys

import pytest

import crafted_sets
import oracle_py


@pytest.fixture(scope="module")
def sets():
    return crafted_sets.all_sets()


def test_crafted_streams_on_the_oracle(sets):
    assert len(sets) > 100
    for name, s, st, exp in sets:
        for flags in (0, oracle_py.FLAG_TREE_WALK):
            got = oracle_py.decode(s, flags, cap=1 << 16)
            if st is not None:
                assert got[0] == st, (name, got[0], st)
            if exp is not None:
                assert got[1] == exp, name


def test_transform_census_covers_all_121_ids():
    """The transform streams really decode as dictionary references with every id 0..120 (oracle trace)."""
    code = ("import sys; sys.path.insert(0, %r); import crafted_sets, oracle_py\n"
            "for n, s, st, e in crafted_sets.transform_streams(): oracle_py.decode(s)\n") % os.path.dirname(__file__)
    env = dict(os.environ, BRO_TRACE="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stderr
    tids = {int(l.split()[1]) for l in out.splitlines() if l.startswith("TID ")}
    assert tids == set(range(121))


def test_omit_first_short_words_follow_the_reference_not_the_spec():
    """Q1: OmitFirstN on a word of length <= N keeps the last byte (src/transformation/mod.rs:89...)."""
    for tid, n in zip(crafted_sets.OMIT_FIRST, (1, 2, 3, 4, 5, 6, 9, 7)):
        pass
    w = b"abcd"
    outs = {tid: oracle_py.transform(tid, w) for tid in crafted_sets.OMIT_FIRST}
    assert sorted(outs.values(), key=len)[0] == b"d"  # never empty
    assert all(len(v) >= 1 for v in outs.values())


def test_many_trees_streams_are_valid_and_switch_block_types():
    """craft.many_trees_stream (round 5: what one piece of > 1 MiB looks like out of an encoder, at test size): valid in both
    prefix-lookup modes, the same bytes in both, with literal and distance block switches and all the trees it asks for."""
    import craft
    for seed, (ntl, ntd, nl, nd) in enumerate([(200, 80, 5, 25), (65, 65, 2, 17), (256, 256, 4, 64), (3, 100, 1, 30), (70, 1, 6, 1), (1, 1, 1, 1)]):
        s = craft.many_trees_stream(seed, ntl, ntd, nl, nd)
        a = oracle_py.decode(s, 0, cap=1 << 16, want_stats=True)
        b = oracle_py.decode(s, oracle_py.FLAG_TREE_WALK, cap=1 << 16)
        assert a[0] == 0 and b[0] == 0 and a[1] == b[1] and len(a[1]) > 3500, (seed, a[0], b[0])
        assert a[2]["commands"] == 400
        if nl > 1 or nd > 1:
            assert a[2]["block_switches"] > 40, (seed, a[2])
    env = dict(os.environ, BRO_TRACE="1")
    code = ("import sys; sys.path.insert(0, %r); import craft, oracle_py\n"
            "oracle_py.decode(craft.many_trees_stream(2, 256, 256, 4, 64), 0, cap=1 << 16)\n") % os.path.dirname(__file__)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stderr
    assert "NTL 256 4 " in out and "NTD 256 64 1" in out, [l for l in out.splitlines() if l.startswith("NT")]
