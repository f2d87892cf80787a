"""GPU parity: the HIP probe path (through the C-ABI of include/txq.h) against the CPU oracle.

Bit-exact comparison on the same seeded inputs — integer/bit work, tolerance zero.
Edge cases follow the domain: empty and ragged batches, 1-word and odd-word rows, rows wider
than one wave sweep, every hash-function count, column shards, all-zero / all-one rows.
"""
import numpy as np
import pytest

from helpers import random_words, oracle_ibf_from_words, random_hibf, layout_hibf, splitmix64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from tetrex_amd import capi as c
    c.init(0)
    return c


def _check_ibf(capi, oracle, bins, bin_size, h, n, seed, density=0.35, shards=(1,)):
    words = random_words(bins, bin_size, density, seed)
    ox = oracle_ibf_from_words(oracle, bins, bin_size, h, words)
    kmers = splitmix64(seed + 1, n) >> np.uint64(int(np.random.default_rng(seed).integers(0, 44)))
    want = ox.probe(kmers)
    W = (bins + 63) // 64
    for R in shards:
        got_cols = []
        for r in range(R):
            ix = capi.Index.upload_ibf(bins, bin_size, h, words, shard_rank=r, n_shards=R)
            lo, nw = int(ix.info.shard_word0), ix.shard_words
            got = ix.probe(kmers)
            assert got.shape == (n, nw)
            assert np.array_equal(got, want[:, lo:lo + nw]), (bins, bin_size, h, n, R, r)
            got_cols.append(got)
            ix.free()
        assert sum(g.shape[1] for g in got_cols) == W


@pytest.mark.parametrize("bins", [5, 64, 65, 128, 200, 300, 1024, 3000, 8192, 9000])
def test_probe_matches_oracle_across_row_widths(capi, oracle, bins):
    _check_ibf(capi, oracle, bins, bin_size=4099, h=3, n=1000, seed=bins)


@pytest.mark.parametrize("h", [1, 2, 3, 4, 5])
def test_probe_every_hash_count(capi, oracle, h):
    for bins in (40, 1024, 777):
        _check_ibf(capi, oracle, bins, bin_size=10007, h=h, n=513, seed=100 + h)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 127, 129, 100003])
def test_probe_ragged_batch_sizes(capi, oracle, n):
    _check_ibf(capi, oracle, 1024, bin_size=2053, h=3, n=n, seed=7 + n)
    _check_ibf(capi, oracle, 50, bin_size=2053, h=2, n=n, seed=9 + n)


@pytest.mark.parametrize("shards", [2, 3, 8, 16, 17])
def test_probe_column_shards_tile_the_full_mask(capi, oracle, shards):
    _check_ibf(capi, oracle, 1024, bin_size=3001, h=3, n=700, seed=shards, shards=(shards,))
    _check_ibf(capi, oracle, 130, bin_size=3001, h=3, n=300, seed=shards + 50, shards=(shards,))


def test_probe_bin_size_extremes(capi, oracle):
    # one row (every k-mer maps to row 0), power-of-two rows, rows not a power of two
    for m in (1, 2, 64, 65, 4096, 1 << 20):
        _check_ibf(capi, oracle, 192, bin_size=m, h=3, n=300, seed=m, density=0.5)


def test_probe_all_zero_and_all_one_matrices(capi, oracle):
    bins, m = 1000, 997
    W = (bins + 63) // 64
    kmers = splitmix64(3, 500)
    zeros = np.zeros(m * W, dtype=np.uint64)
    ix = capi.Index.upload_ibf(bins, m, 3, zeros)
    assert not ix.probe(kmers).any()
    ix.free()
    ones = random_words(bins, m, 1.1, 0)  # density > 1: every real bin set, padding bits clear
    ix = capi.Index.upload_ibf(bins, m, 3, ones)
    got = ix.probe(kmers)
    want = oracle_ibf_from_words(oracle, bins, m, 3, ones).probe(kmers)
    assert np.array_equal(got, want)
    assert int(got[0, -1]) == (1 << (bins % 64)) - 1  # bits >= bins stay zero
    ix.free()


def test_alive_bits_match_mask_nonzero(capi, oracle):
    for bins, dens in ((1024, 0.1), (40, 0.26), (3000, 0.07), (9000, 0.048)):
        m, n = 1553, 1000
        words = random_words(bins, m, dens, bins)
        ix = capi.Index.upload_ibf(bins, m, 3, words)
        kmers = splitmix64(11, n)
        dk = capi.DeviceBuffer.from_numpy(kmers)
        dm = capi.DeviceBuffer(n * ix.shard_words * 8)
        da = capi.DeviceBuffer(((n + 63) // 64) * 8)
        ix.probe_device(dk.ptr, n, dm.ptr, da.ptr)
        capi.synchronize()
        masks = dm.to_numpy(np.uint64, (n, ix.shard_words))
        alive = da.to_numpy(np.uint64, ((n + 63) // 64,))
        want = oracle_ibf_from_words(oracle, bins, m, 3, words).probe(kmers)
        assert np.array_equal(masks, want)
        bits = np.unpackbits(alive.view(np.uint8), bitorder="little")[:n].astype(bool)
        assert np.array_equal(bits, want.any(axis=1))
        assert 0 < bits.sum() < n  # the test exercises both outcomes
        ix.free()


def test_device_emplace_builds_the_oracle_matrix(capi, oracle):
    for bins, m, h in ((5, 106, 3), (1024, 5003, 3), (300, 999, 2), (64, 64, 3)):
        n = 20000
        rng = np.random.default_rng(bins)
        values = rng.integers(0, 1 << 20, size=n, dtype=np.uint64)
        bins_of = rng.integers(0, bins, size=n, dtype=np.uint32)
        ox = oracle.Index.ibf(bins, m, h, dna=False, k=4)
        ox.emplace_pairs(values, bins_of)
        ix = capi.Index.create_ibf(bins, m, h)
        dv = capi.DeviceBuffer.from_numpy(values)
        db = capi.DeviceBuffer.from_numpy(bins_of)
        ix.emplace_device(dv.ptr, db.ptr, n)
        capi.synchronize()
        assert np.array_equal(ix.download_words_rows(m), ox.words())
        # no false negatives: every inserted value reports its bin
        got = ix.probe(values[:2000])
        b = bins_of[:2000].astype(np.int64)
        assert np.all((got[np.arange(2000), b >> 6] >> (b & 63).astype(np.uint64)) & np.uint64(1))
        ix.free()


def test_fixture_bits_on_gpu(capi, oracle, golden):
    """The reference-built fixture probed on the GPU: M[ACG]=M[ACC]=0b11, M[CCG]=0b01."""
    import os
    from conftest import GOLDEN
    fx = oracle.read_legacy_fixture(os.path.join(GOLDEN, "ibf_idx.ibf"))
    ix = capi.Index.upload_ibf(2, 64, 3, fx["words"])
    got = ix.probe([7, 5, 23])
    assert [int(x) for x in got[:, 0]] == [0b11, 0b11, 0b01]
    ix.free()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_hibf_descent_matches_oracle(capi, oracle, seed):
    ox, descs, values = random_hibf(oracle, seed, user_bins=300, levels=3)
    ix = capi.Index.upload_hibf(300, descs)
    assert ix.info.is_hibf == 1 and ix.info.n_ibf == len(descs)
    rng = np.random.default_rng(seed)
    present = np.concatenate([v[:5] for v in values])
    absent = rng.integers(0, 1 << 20, size=3000, dtype=np.uint64)
    kmers = np.concatenate([present, absent])
    got = ix.probe(kmers)
    want = ox.probe(kmers)
    assert np.array_equal(got, want)
    # no false negatives for inserted values
    for ub in range(300):
        for j in range(5):
            assert (int(got[ub * 5 + j, ub >> 6]) >> (ub & 63)) & 1
    ix.free()


def test_hibf_sharded_masks_and_wide_tree(capi, oracle):
    ox, descs, values = random_hibf(oracle, 9, user_bins=1500, tmax=512, levels=4, n_values=10)
    kmers = np.concatenate([np.concatenate([v[:2] for v in values]), splitmix64(5, 2000) >> np.uint64(44)])
    want = ox.probe(kmers)
    for R in (1, 3):
        for r in range(R):
            ix = capi.Index.upload_hibf(1500, descs, shard_rank=r, n_shards=R)
            lo, nw = int(ix.info.shard_word0), ix.shard_words
            assert np.array_equal(ix.probe(kmers), want[:, lo:lo + nw])
            ix.free()


@pytest.mark.parametrize("shape", [dict(user_bins=40, tmax=32, levels=2), dict(user_bins=300, tmax=128, levels=3),
                                   dict(user_bins=2500, tmax=2048, levels=3, n_values=8), dict(user_bins=700, tmax=200, levels=5, n_values=12)])
def test_hibf_fused_and_level_synchronous_kernels_agree(capi, oracle, shape, monkeypatch):
    """The same tree through all descent kernels (txq_hibf.hip: hibf_small_kernel for small trees and
    hibf_fused_kernel otherwise are the default, TXQ_HIBF_SMALL=0 keeps small trees on the fused kernel,
    TXQ_HIBF_LEVELS=1 selects hibf_level_kernel): masks and alive bits equal the oracle's, whatever
    the row width (one word, a lane's four words, several lanes), the depth and the shard."""
    ox, descs, values = random_hibf(oracle, 31, **shape)
    ub = shape["user_bins"]
    kmers = np.concatenate([np.concatenate([v[:3] for v in values]), splitmix64(8, 1500) >> np.uint64(44)])
    want = ox.probe(kmers)
    for R, r in ((1, 0), (2, 1)):
        ix = capi.Index.upload_hibf(ub, descs, shard_rank=r, n_shards=R)
        lo, nw = int(ix.info.shard_word0), ix.shard_words
        dk = capi.DeviceBuffer.from_numpy(kmers)
        # small kernel (lane per k-mer, where the tree qualifies) / wave-per-k-mer kernel / level-synchronous
        for levels, small in (("0", "1"), ("0", "0"), ("1", "1")):
            monkeypatch.setenv("TXQ_HIBF_LEVELS", levels)
            monkeypatch.setenv("TXQ_HIBF_SMALL", small)
            dm = capi.DeviceBuffer(kmers.size * nw * 8)
            da = capi.DeviceBuffer(((kmers.size + 63) // 64) * 8)
            ix.probe_device(dk.ptr, kmers.size, dm.ptr, da.ptr)
            capi.synchronize()
            got = dm.to_numpy(np.uint64, (kmers.size, nw))
            assert np.array_equal(got, want[:, lo:lo + nw]), (shape, R, levels, small)
            alive = np.unpackbits(da.to_numpy(np.uint8, (((kmers.size + 63) // 64) * 8,)), bitorder="little")[:kmers.size]
            assert np.array_equal(alive.astype(bool), got.any(axis=1)), (shape, R, levels, small)
        ix.free()


def test_host_buffer_probe_pipelines_chunks_into_pageable_and_pinned_memory(capi, oracle):
    """txq_probe (host buffers) cuts the batch into chunks and overlaps probe and copy-back; the
    result is the same in pageable and in pinned (txq_host_alloc) output memory, for a flat IBF
    (3 chunks) and an HIBF (2 chunks), and equals the oracle on a sample."""
    bins, m = 1024, 50021
    words = random_words(bins, m, 0.3, 77)
    ix = capi.Index.upload_ibf(bins, m, 3, words)
    n = 700000
    kmers = splitmix64(21, n) >> np.uint64(40)
    got = ix.probe(kmers)
    pinned = capi.HostBuffer((n, 16))
    ix.probe(kmers, out=pinned.array)
    assert np.array_equal(got, pinned.array)
    dk = capi.DeviceBuffer.from_numpy(kmers)
    dm = capi.DeviceBuffer(n * 16 * 8)
    ix.probe_device(dk.ptr, n, dm.ptr)
    capi.synchronize()
    assert np.array_equal(got, dm.to_numpy(np.uint64, (n, 16)))
    idx = np.arange(0, n, 997)
    assert np.array_equal(got[idx], oracle_ibf_from_words(oracle, bins, m, 3, words).probe(kmers[idx]))
    pinned.free()
    ix.free()

    ox, descs, values = random_hibf(oracle, 5, user_bins=300, levels=3)
    hx = capi.Index.upload_hibf(300, descs)
    n = (1 << 20) + 12345
    kmers = np.resize(np.concatenate([np.concatenate([v[:4] for v in values]), splitmix64(9, 5000) >> np.uint64(44)]), n)
    got = hx.probe(kmers)
    dk = capi.DeviceBuffer.from_numpy(kmers)
    dm = capi.DeviceBuffer(n * hx.shard_words * 8)
    hx.probe_device(dk.ptr, n, dm.ptr)
    capi.synchronize()
    assert np.array_equal(got, dm.to_numpy(np.uint64, (n, hx.shard_words)))
    assert np.array_equal(got[:6200], ox.probe(kmers[:6200]))
    hx.free()


def test_hibf_with_a_different_hash_count_per_ibf(capi, oracle, monkeypatch):
    """The reference's HIBF uses one hash count for the whole tree, the descriptor allows one per IBF:
    root h = 3, children h = 1 / 2 / 3 / 4.  An IBF with fewer hash functions than the tree's maximum
    repeats its last row in the kernels; all three descent kernels must agree with the oracle."""
    from helpers import MERGED
    rng = np.random.default_rng(12)
    hs, per_child, user_bins = [1, 2, 3, 4], 70, 280
    vals = [rng.integers(0, 1 << 20, size=30, dtype=np.uint64) for _ in range(user_bins)]
    ox = oracle.Index.hibf(user_bins, dna=False, k=4)
    nxt, tbu = np.arange(1, 5, dtype=np.uint64), np.full(4, MERGED, dtype=np.uint64)
    descs = [dict(bins=4, bin_size=20011, hash_funs=3, next_ibf_id=nxt, tb_to_user=tbu)]
    ox.add_ibf(4, 20011, 3, nxt, tbu)
    for c, h in enumerate(hs):
        ox.hibf_emplace(0, np.concatenate(vals[c * per_child:(c + 1) * per_child]), c)
    for c, h in enumerate(hs):
        tb = np.arange(c * per_child, (c + 1) * per_child, dtype=np.uint64)
        i = ox.add_ibf(per_child, 1009, h, np.zeros(per_child, dtype=np.uint64), tb)
        for t in range(per_child):
            ox.hibf_emplace(i, vals[c * per_child + t], t)
        descs.append(dict(bins=per_child, bin_size=1009, hash_funs=h, next_ibf_id=np.zeros(per_child, dtype=np.uint64), tb_to_user=tb))
    for i, d in enumerate(descs):
        d["words"] = ox.hibf_words(i)
    kmers = np.concatenate([np.concatenate([v[:3] for v in vals]), splitmix64(14, 3000) >> np.uint64(44)])
    want = ox.probe(kmers)
    ix = capi.Index.upload_hibf(user_bins, descs)
    for levels, small in (("0", "1"), ("0", "0"), ("1", "1")):
        monkeypatch.setenv("TXQ_HIBF_LEVELS", levels)
        monkeypatch.setenv("TXQ_HIBF_SMALL", small)
        assert np.array_equal(ix.probe(kmers), want), (levels, small)
    ix.free()


def test_full_size_swissprot_shape_properties(capi, oracle):
    """BASELINE configs[1] shape (1024 bins, h=3, m=1,247,045 rows, 160 MB): size-independent
    properties at full size, and a sampled bit-exact comparison against the oracle."""
    bins, m, h = 1024, oracle.compute_bitcount(200000, 0.05), 3
    assert m == 1247045
    ix = capi.Index.create_ibf(bins, m, h)
    n_ins = 1 << 22
    values = splitmix64(1, n_ins) >> np.uint64(44)          # uniform 20-bit k-mers (k=4, 5 bits/residue)
    bins_of = (splitmix64(2, n_ins) % np.uint64(bins)).astype(np.uint32)
    dv, db = capi.DeviceBuffer.from_numpy(values), capi.DeviceBuffer.from_numpy(bins_of)
    ix.emplace_device(dv.ptr, db.ptr, n_ins)
    capi.synchronize()
    # (1) no false negatives, (2) idempotence of insertion, (3) probe determinism
    q = values[:50000]
    g1 = ix.probe(q)
    b = bins_of[:50000].astype(np.int64)
    assert np.all((g1[np.arange(q.size), b >> 6] >> (b & 63).astype(np.uint64)) & np.uint64(1))
    ix.emplace_device(dv.ptr, db.ptr, n_ins)
    capi.synchronize()
    assert np.array_equal(ix.probe(q), g1)
    # (4) sampled parity: the oracle gets the device-built matrix and must agree on fresh k-mers
    words = ix.download_words_rows(m)
    ox = oracle_ibf_from_words(oracle, bins, m, h, words)
    fresh = splitmix64(3, 20000) >> np.uint64(44)
    assert np.array_equal(ix.probe(fresh), ox.probe(fresh))
    # (5) the matrix itself equals an oracle build of a slice of the insertions? -> checksum of rows
    sub = oracle.Index.ibf(bins, m, h, dna=False, k=4)
    sub.emplace_pairs(values[:200000], bins_of[:200000])
    w_sub = sub.words()
    assert np.array_equal(w_sub & words, w_sub)  # every bit the oracle sets is set on the device
    ix.free()


def test_shards_without_mask_words_and_empty_batches(capi, oracle):
    """More shards than mask words: a rank that owns no column answers with zero-width masks (probes and
    whole queries, flat and hierarchical); empty batches are no-ops."""
    w = random_words(5, 101, 0.3, 1)
    for r, width in ((0, 1), (1, 0)):
        ix = capi.Index.upload_ibf(5, 101, 3, w, shard_rank=r, n_shards=2)
        assert ix.shard_words == width and ix.probe(splitmix64(1, 100) >> np.uint64(58)).shape == (100, width)
        masks, status, _ = ix.query_masks(["ACG", "A.T"], True, 3)
        assert masks.shape == (2, width) and status == [0, 0]
        ix.free()
    ox, descs, values = random_hibf(oracle, 3, user_bins=40, tmax=32, levels=2)
    kmers = np.concatenate([v[:2] for v in values])
    for r, width in ((0, 1), (1, 0), (2, 0)):
        hx = capi.Index.upload_hibf(40, descs, shard_rank=r, n_shards=3)
        got = hx.probe(kmers)
        assert got.shape == (kmers.size, width)
        if width:
            assert np.array_equal(got, ox.probe(kmers))
        masks, status, _ = hx.query_masks(["LMAEG"], False, 4)
        assert masks.shape == (1, width) and status == [0]
        hx.free()
    ix = capi.Index.upload_ibf(5, 101, 3, w)
    assert ix.probe(np.zeros(0, dtype=np.uint64)).shape == (0, 1) and ix.query_masks([], True, 3)[0].shape == (0, 1)
    ix.free()


@pytest.mark.parametrize("shape", [dict(user_bins=128 * 70, children=70, h=2), dict(user_bins=256 * 33, children=33, h=3),
                                   dict(user_bins=512 * 9, children=9, h=1), dict(user_bins=65536, children=256, h=2)],
                         ids=["70x128", "33x256", "9x512", "256x256"])
def test_hibf_child_stationary_kernel_on_regular_two_level_trees(capi, oracle, shape, monkeypatch):
    """Regular two-level trees (the layout `tetrex index` writes) take the child-stationary descent
    (txq_hibf.hip: hibf_root_kernel + hibf_children_kernel): row widths of 2, 4 and 8 words per child, child
    counts that do not fill the last wave step, 1-3 hash functions, column shards, ragged batch sizes, alive
    bits — against the oracle's membership_for restatement, and against the k-mer-stationary kernel
    (TXQ_HIBF_STATIONARY=0) on the same device buffers."""
    from helpers import regular_hibf
    ub, ch, h = shape["user_bins"], shape["children"], shape["h"]
    rng = np.random.default_rng(ub)
    per = 12 if ub > 20000 else 30
    ox, descs, values = regular_hibf(oracle, ub, ch, per, lambda b: rng.integers(0, 1 << 20, size=per, dtype=np.uint64), h=h)
    present = np.concatenate([values[b][:1] for b in range(0, ub, 37)])
    kmers = np.concatenate([present, splitmix64(8, 1531) >> np.uint64(44)])
    want = ox.probe(kmers)
    assert want.any()
    for R, r in ((1, 0), (4, 2)) if ch % 4 == 0 else ((1, 0),):
        ix = capi.Index.upload_hibf(ub, descs, shard_rank=r, n_shards=R)
        lo, nw = int(ix.info.shard_word0), ix.shard_words
        dk = capi.DeviceBuffer.from_numpy(kmers)
        results = []
        for stationary in ("1", "0"):
            monkeypatch.setenv("TXQ_HIBF_STATIONARY", stationary)
            dm = capi.DeviceBuffer(kmers.size * nw * 8)
            da = capi.DeviceBuffer(((kmers.size + 63) // 64) * 8)
            ix.probe_device(dk.ptr, kmers.size, dm.ptr, da.ptr)
            capi.synchronize()
            got = dm.to_numpy(np.uint64, (kmers.size, nw))
            assert np.array_equal(got, want[:, lo:lo + nw]), (shape, R, stationary)
            alive = np.unpackbits(da.to_numpy(np.uint8, (((kmers.size + 63) // 64) * 8,)), bitorder="little")[:kmers.size]
            assert np.array_equal(alive.astype(bool), got.any(axis=1)), (shape, R, stationary)
            results.append(got)
        assert np.array_equal(results[0], results[1])
        for n in (1, 63, 257):  # ragged batches through the host-buffer entry point
            assert np.array_equal(ix.probe(kmers[:n]), want[:n, lo:lo + nw])
        ix.free()


@pytest.mark.parametrize("shape", [dict(user_bins=1024, children=16, h=3), dict(user_bins=6 * 256, children=6, h=2), dict(user_bins=3 * 64 - 9, children=3, h=1),
                                   dict(user_bins=32 * 64, children=32, h=2), dict(user_bins=5 * 128, children=5, h=4)],
                         ids=["16x64", "6x256", "3x64-ragged", "32x64", "5x128"])
def test_small_uniform_trees_are_probed_on_their_interleaved_children(capi, oracle, shape, monkeypatch):
    """A regular two-level tree with a root of at most 64 merged bins, a mask of at most 32 words and children of equal rows
    and hash counts (what `tetrex index` writes for up to 2048 bins) keeps its children once more side by side, and a
    plain k-mer probe gathers them like a flat IBF; the root's word clears the children the k-mer cannot be in
    (txq_probe.hip TreeRoot).  Masks and alive bits against the oracle's membership_for restatement and against the
    descent kernels (TXQ_HIBF_INTERLEAVE_PROBE=0) on the same buffers; column shards; ragged batches."""
    from helpers import regular_hibf
    ub, ch, h = shape["user_bins"], shape["children"], shape["h"]
    rng = np.random.default_rng(ub + h)
    per = 40
    ox, descs, values = regular_hibf(oracle, ub, ch, per, lambda b: rng.integers(0, 1 << 20, size=per, dtype=np.uint64), h=h)
    rows = max(d["bin_size"] for d in descs[1:])
    if len({d["bin_size"] for d in descs[1:]}) > 1:  # (regular_hibf sizes a child by its own largest bin: make them equal, as the product does)
        pytest.skip("children of different rows")
    present = np.concatenate([values[b][:2] for b in range(0, ub, 7)])
    kmers = np.concatenate([present, splitmix64(8, 1531) >> np.uint64(44)])
    want = ox.probe(kmers)
    assert want.any()
    for R, r in ((1, 0), (2, 1)) if ch % 2 == 0 else ((1, 0),):
        ix = capi.Index.upload_hibf(ub, descs, shard_rank=r, n_shards=R)
        lo, nw = int(ix.info.shard_word0), ix.shard_words
        dk = capi.DeviceBuffer.from_numpy(kmers)
        results = []
        for interleaved in ("1", "0"):
            monkeypatch.setenv("TXQ_HIBF_INTERLEAVE_PROBE", interleaved)
            dm = capi.DeviceBuffer(kmers.size * nw * 8)
            da = capi.DeviceBuffer(((kmers.size + 63) // 64) * 8)
            ix.probe_device(dk.ptr, kmers.size, dm.ptr, da.ptr)
            capi.synchronize()
            got = dm.to_numpy(np.uint64, (kmers.size, nw))
            assert np.array_equal(got, want[:, lo:lo + nw]), (shape, R, interleaved)
            alive = np.unpackbits(da.to_numpy(np.uint8, (((kmers.size + 63) // 64) * 8,)), bitorder="little")[:kmers.size]
            assert np.array_equal(alive.astype(bool), got.any(axis=1)), (shape, R, interleaved)
            results.append(got)
        assert np.array_equal(results[0], results[1])
        monkeypatch.delenv("TXQ_HIBF_INTERLEAVE_PROBE")
        for n in (1, 63, 257):
            assert np.array_equal(ix.probe(kmers[:n]), want[:n, lo:lo + nw])
        ix.free()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["layout-16", "layout-64", "layout-32"])
def test_hibf_stack_entries_beyond_the_lds_go_through_the_output_row(capi, oracle, shape, monkeypatch):
    """hibf_fused_kernel keeps the first TXQ_HIBF_STACK_LDS entries (default 128) of a k-mer's stack of pending IBFs in LDS and
    the rest in the k-mer's own output row until the finished row is written over them.  Trees whose bins draw their values
    from a universe of 64: a k-mer reaches nearly every IBF, dozens to hundreds are pending at once.  Masks of plain probes
    (user order) and of a session's rows of plain k-mers (layout order, through a literal query per k-mer) equal the oracle's
    with 2, 32, 96 and the default number of entries in LDS (where the output row is too short for the rest — 2 * w_out entries;
    user order on the narrower masks here — the whole stack stays in LDS)."""
    if shape == "layout-16":
        ox, descs, values = layout_hibf(oracle, 3, user_bins=1500, tmax=16, n_values=30, value_bits=6, direct=3)
        ub = 1500
    elif shape == "layout-64":
        ox, descs, values = layout_hibf(oracle, 4, user_bins=6000, tmax=64, n_values=20, value_bits=6)
        ub = 6000
    else:
        ox, descs, values = layout_hibf(oracle, 8, user_bins=4000, tmax=32, n_values=20, value_bits=6, h=3)
        ub = 4000
    assert len(descs) > 100  # (IBFs: more than the stacks' LDS parts hold)
    kmers = np.concatenate([np.arange(64, dtype=np.uint64), splitmix64(3, 500) >> np.uint64(44)])
    want = ox.probe(kmers)
    assert (np.unpackbits(want[:64].view(np.uint8), axis=1).sum(axis=1) > ub // 4).any()  # saturated: some k-mers are nearly everywhere
    ix = capi.Index.upload_hibf(ub, descs)
    monkeypatch.setenv("TXQ_HIBF_SMALL", "0")
    for in_lds in ("2", "32", "96", None):
        if in_lds:  # (layout order with the row in LDS: writing the row directly, the default, leaves the LDS to the whole stack)
            monkeypatch.setenv("TXQ_HIBF_STACK_LDS", in_lds)
            monkeypatch.setenv("TXQ_HIBF_LAYOUT_DIRECT", "0")
        else:
            monkeypatch.delenv("TXQ_HIBF_STACK_LDS", raising=False)
            monkeypatch.delenv("TXQ_HIBF_LAYOUT_DIRECT", raising=False)
        assert np.array_equal(ix.probe(kmers), want), (shape, in_lds)
        # peptide k = 4 literals: their masks are the rows of their one k-mer (layout order where the index has one)
        qs = ["ACDE", "AAAA", "AAAC", "AACA"]
        got, status, _ = ix.query_masks(qs, False, 4, 0, 0)
        for q, g, st in zip(qs, got, status):
            w, _ = ox.expected_mask(q)
            assert st == 0 and np.array_equal(g, w), (shape, in_lds, q)
    ix.free()
