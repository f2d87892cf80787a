"""GPU parity of the mask-DAG executor (txq_run_programs) against a numpy evaluation of the
same programs over oracle-probed masks.  Bit-exact."""
import numpy as np
import pytest

from helpers import (random_words, oracle_ibf_from_words, random_hibf, splitmix64, make_blob, eval_program,
                     ones_mask, NO_KMER)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from tetrex_amd import capi as c
    c.init(0)
    return c


def _random_program(rng, n_kmers, n_slots, n_ops):
    """Random but well-formed: every slot is written before it is read."""
    written = [0, 1, 2]
    ops = []
    for _ in range(n_ops):
        a = int(rng.choice(written))
        b = int(rng.choice(written)) if rng.random() < 0.5 else 0
        k = int(rng.integers(0, n_kmers)) if rng.random() < 0.7 else NO_KMER
        d = int(rng.integers(2, n_slots))
        ops.append((k, d, a, b))
        if d not in written:
            written.append(d)
    ops.append((NO_KMER, 2, int(rng.choice(written)), 2))  # Match: RESULT |= slot
    return n_slots, ops


@pytest.mark.parametrize("bins", [5, 64, 130, 1024, 3000, 9000])
def test_random_programs_match_numpy(capi, oracle, bins):
    rng = np.random.default_rng(bins)
    m, h = 2053, 3
    words = random_words(bins, m, 0.6, bins)
    ox = oracle_ibf_from_words(oracle, bins, m, h, words)
    kmers = splitmix64(bins, 200) >> np.uint64(40)
    M = ox.probe(kmers)
    programs = [_random_program(rng, kmers.size, int(rng.integers(3, 40)), int(rng.integers(0, 120))) for _ in range(257)]
    blob = make_blob(kmers, programs)
    for R in (1, 3):
        for r in range(R):
            ix = capi.Index.upload_ibf(bins, m, h, words, shard_rank=r, n_shards=R)
            lo, nw = int(ix.info.shard_word0), ix.shard_words
            got = ix.run_programs(blob, len(programs))
            ones = ones_mask(bins, lo, nw)
            for p, (n_slots, ops) in enumerate(programs):
                want = eval_program(n_slots, ops, M[:, lo:lo + nw], ones)
                assert np.array_equal(got[p], want), (bins, R, r, p)
            ix.free()


def test_program_without_probes_returns_all_real_bins(capi, oracle):
    """A query whose paths are all shorter than k never probes: the result is
    hit_vector(bin_count, true) (include/otf_collector.h:344,361)."""
    bins, m = 1000, 101
    words = random_words(bins, m, 0.5, 1)
    ix = capi.Index.upload_ibf(bins, m, 3, words)
    blob = make_blob(np.zeros(0, dtype=np.uint64), [(3, [(NO_KMER, 2, 1, 2)]), (3, [])])
    got = ix.run_programs(blob, 2)
    assert np.array_equal(got[0], ones_mask(bins))
    assert not got[1].any()
    ix.free()


def test_config2_motif_program_by_hand(capi, oracle, golden):
    """LMA(E|Q)GLYN, k=4: (M[LMAE]&M[MAEG]&M[AEGL]&M[EGLY] | M[LMAQ]&M[MAQG]&M[AQGL]&M[QGLY]) & M[GLYN]
     — the oracle's collector against a hand-written program on the GPU."""
    g = golden("config2_kmers.json")["kmers"]
    bins, m, h = 1024, 30011, 3
    ox = oracle.Index.ibf(bins, m, h, dna=False, k=4)
    rng = np.random.default_rng(5)
    names = list(g)
    for b in range(bins):
        keep = [g[n] for n in names if rng.random() < 0.8]
        if keep:
            ox.emplace(keep, b)
        ox.emplace(rng.integers(0, 1 << 20, size=300, dtype=np.uint64), b)
    want = ox.query("LMA(E|Q)GLYN")
    idx = {n: i for i, n in enumerate(names)}
    ops = [(idx["LMAE"], 3, 1, 0), (idx["MAEG"], 3, 3, 0), (idx["AEGL"], 3, 3, 0), (idx["EGLY"], 3, 3, 0),
           (idx["LMAQ"], 4, 1, 0), (idx["MAQG"], 4, 4, 0), (idx["AQGL"], 4, 4, 0), (idx["QGLY"], 4, 4, 0),
           (NO_KMER, 3, 3, 4), (idx["GLYN"], 3, 3, 0), (NO_KMER, 2, 3, 2)]
    blob = make_blob(np.array([g[n] for n in names], dtype=np.uint64), [(5, ops)])
    ix = capi.Index.upload_ibf(bins, m, h, ox.words())
    got = ix.run_programs(blob, 1)[0]
    assert np.array_equal(got, want)
    assert 0 < int(np.unpackbits(got.view(np.uint8)).sum()) < bins
    ix.free()


def test_programs_on_hibf(capi, oracle):
    ox, descs, values = random_hibf(oracle, 4, user_bins=300, levels=3)
    rng = np.random.default_rng(4)
    kmers = np.concatenate([np.concatenate([v[:1] for v in values[:150]]), rng.integers(0, 1 << 20, 50, dtype=np.uint64)])
    M = ox.probe(kmers)
    programs = [_random_program(rng, kmers.size, 12, 40) for _ in range(64)]
    blob = make_blob(kmers, programs)
    ix = capi.Index.upload_hibf(300, descs)
    got = ix.run_programs(blob, len(programs))
    ones = ones_mask(300)
    for p, (n_slots, ops) in enumerate(programs):
        assert np.array_equal(got[p], eval_program(n_slots, ops, M, ones)), p
    ix.free()


def test_malformed_blobs_are_rejected_on_the_host(capi, oracle):
    ix = capi.Index.upload_ibf(64, 8, 2, np.zeros(8, dtype=np.uint64))
    bad = [
        make_blob(np.zeros(1, dtype=np.uint64), [(3, [(0, 5, 1, 0)])]),        # dst >= n_slots
        make_blob(np.zeros(1, dtype=np.uint64), [(4, [(7, 3, 1, 0)])]),        # k-mer index out of range
        make_blob(np.zeros(1, dtype=np.uint64), [(4, [(0, 1, 1, 0)])]),        # writes ONES
        make_blob(np.zeros(1, dtype=np.uint64), [(2, [])]),                    # n_slots < 3
        b"\0" * 64,                                                             # bad magic
    ]
    for blob in bad:
        with pytest.raises(capi.TxqError) as e:
            ix.run_programs(blob, 1)
        assert e.value.code == -6
    ix.free()
