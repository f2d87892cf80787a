"""CPU tests of the bin-shard join (host/compiler.hpp join_shard_masks — the 'OR-reduce' of a bin-sharded index is a
column concatenation because the shards are disjoint) and of the shard ranges libtxq uses."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def host():
    from tetrex_amd import host as H
    H.lib()
    return H


def _ranges(words, R):
    base, rem = divmod(words, R)
    lo = [base * r + min(r, rem) for r in range(R)]
    return lo, [base + (1 if r < rem else 0) for r in range(R)]


@pytest.mark.parametrize("words,R", [(16, 1), (16, 8), (17, 3), (128, 8), (5, 8), (1024, 6)])
def test_join_reassembles_the_full_masks(host, words, R):
    rng = np.random.default_rng(words * 31 + R)
    full = rng.integers(0, 1 << 63, size=(37, words), dtype=np.uint64)
    lo, n = _ranges(words, R)
    parts = [full[:, lo[r]:lo[r] + n[r]].copy() for r in range(R)]
    order = rng.permutation(R)  # the join does not depend on the order the shards are listed in
    got = host.join_shard_masks(words, [lo[r] for r in order], [parts[r] for r in order])
    assert np.array_equal(got, full)


def test_join_refuses_gaps_overlaps_and_overruns(host):
    a = np.zeros((3, 4), dtype=np.uint64)
    with pytest.raises(host.HostError):
        host.join_shard_masks(10, [0, 4], [a, a])          # words 8, 9 missing
    with pytest.raises(host.HostError):
        host.join_shard_masks(8, [0, 3], [a, a])           # overlap (and a gap)
    with pytest.raises(host.HostError):
        host.join_shard_masks(8, [0, 6], [a, a])           # runs past the mask
    assert host.join_shard_masks(8, [4, 0], [a, a]).shape == (3, 8)


def test_join_of_zero_queries(host):
    a = np.zeros((0, 4), dtype=np.uint64)
    assert host.join_shard_masks(8, [0, 4], [a, a]).shape == (0, 8)
