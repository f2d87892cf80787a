"""Maximum sizes the reference allows on this path: k = 12 for peptides (60-bit k-mers,
src/main.cpp:28), k = 32 for DNA (64-bit k-mers, include/nucleotide_decomposer.h:36), and rows far
wider than one wave sweep (tens of thousands of bins in a flat IBF).  GPU vs oracle, bit-exact."""
import numpy as np
import pytest

from helpers import random_words, oracle_ibf_from_words, splitmix64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from tetrex_amd import capi as c
    c.init(0)
    return c


def test_seventy_thousand_bins_in_one_flat_ibf(capi, oracle):
    bins, m, h = 70001, 257, 2
    words = random_words(bins, m, 0.5, 1)
    ox = oracle_ibf_from_words(oracle, bins, m, h, words)
    kmers = splitmix64(2, 300)
    for R in (1, 3):
        for r in range(R):
            ix = capi.Index.upload_ibf(bins, m, h, words, shard_rank=r, n_shards=R)
            lo, nw = int(ix.info.shard_word0), ix.shard_words
            assert np.array_equal(ix.probe(kmers), ox.probe(kmers)[:, lo:lo + nw])
            ix.free()
    ix = capi.Index.upload_ibf(bins, m, h, words)
    got, status, _ = ix.query_masks(["LMAEG", "A.CD", "K[RK]DE"], False, 4)
    for rx, g in zip(["LMAEG", "A.CD", "K[RK]DE"], got):
        assert np.array_equal(g, ox.query(rx)), rx
    ix.free()


def _planted_index(oracle, dna, k, bins, seqs):
    per_bin = [oracle.decompose(s, k, dna=dna) for s in seqs]
    m = oracle.compute_bitcount(max(len(v) for v in per_bin), 0.05)
    ox = oracle.Index.ibf(bins, m, 3, dna=dna, k=k)
    for b, v in enumerate(per_bin):
        ox.emplace(v, b)
    return ox, m


def test_peptide_k12(capi, oracle):
    rng = np.random.default_rng(3)
    aa = list("ACDEFGHIKLMNPQRSTVWY")
    bins = 130
    seqs = ["".join(rng.choice(aa, size=300)) for _ in range(bins)]
    motif = "LMAEGLYNKRDEWWHHPQ"
    for b in (7, 99):
        seqs[b] = seqs[b][:50] + motif + seqs[b][68:]
    ox, m = _planted_index(oracle, False, 12, bins, seqs)
    ix = capi.Index.upload_ibf(bins, m, 3, ox.words())
    qs = [motif, motif[:12], motif[:11], "LMAEGLYN(K|R)RDEWWHH", "LMAEGLYNKRDE.WHHPQ", "LMAEGLYNKR[DE]{2}WWHHPQ"]
    got, status, _ = ix.query_masks(qs, False, 12)
    assert not any(status)
    for rx, g in zip(qs, got):
        want, st = ox.query(rx, with_stats=True)
        if st["quirk_merges"] == 0:
            assert np.array_equal(g, want), rx
    assert [b for b in range(bins) if (int(got[0][b >> 6]) >> (b & 63)) & 1] == [7, 99]
    # 60-bit k-mer values go through the hash unchanged
    v = np.array(oracle.decompose(motif, 12, dna=False), dtype=np.uint64)
    assert int(v.max()) >> 55
    assert np.array_equal(ix.probe(v), ox.probe(v))
    ix.free()


def test_dna_k32(capi, oracle):
    rng = np.random.default_rng(4)
    bins = 70
    seqs = ["".join(rng.choice(list("ACGT"), size=400)) for _ in range(bins)]
    motif = "ACGTTGCAAGGCTTAACCGGATATCGCGTATGCAAT"  # 36 nt
    seqs[11] = seqs[11][:100] + motif + seqs[11][136:]
    rc = motif[::-1].translate(str.maketrans("ACGT", "TGCA"))
    seqs[42] = seqs[42][:30] + rc + seqs[42][66:]  # reverse strand: canonical k-mers match too
    ox, m = _planted_index(oracle, True, 32, bins, seqs)
    ix = capi.Index.upload_ibf(bins, m, 3, ox.words())
    qs = [motif, motif[:32], "ACGTTGCAAGGCTTAACCGGATATCGCGTATGC(A|G)AT", motif[:20] + "." + motif[21:]]
    got, status, _ = ix.query_masks(qs, True, 32)
    assert not any(status)
    for rx, g in zip(qs, got):
        want, st = ox.query(rx, with_stats=True)
        if st["quirk_merges"] == 0:
            assert np.array_equal(g, want), rx
    hits = [b for b in range(bins) if (int(got[0][b >> 6]) >> (b & 63)) & 1]
    assert 11 in hits and 42 in hits
    v = np.array(oracle.decompose(motif, 32, dna=True), dtype=np.uint64)
    assert np.array_equal(ix.probe(v), ox.probe(v))
    ix.free()


def test_full_size_8192_bin_shard_properties(capi):
    """BASELINE configs[3] at full size, one GPU's share: the 8192-bin, 62.5M-row DNA IBF (64 GB) sharded
    8 ways is 1024 bin columns x 62.5M rows = 8 GB per rank.  Size-independent properties on that shard
    (rank 3 of 8): no false negatives, bins of other shards are ignored, idempotent insertion,
    deterministic probes, and row-level agreement with a second, independently filled copy."""
    bins, m, h = 8192, 62_500_000, 3
    ix = capi.Index.create_ibf(bins, m, h, shard_rank=3, n_shards=8)
    assert ix.shard_words == 16 and int(ix.info.shard_word0) == 48 and int(ix.info.device_bytes) == 8_000_000_000
    n = 1 << 22
    values = splitmix64(11, n) >> np.uint64(24)                      # 40-bit values: rows spread over all 62.5M
    bins_of = (splitmix64(12, n) % np.uint64(bins)).astype(np.uint32)
    dv, db = capi.DeviceBuffer.from_numpy(values), capi.DeviceBuffer.from_numpy(bins_of)
    ix.emplace_device(dv.ptr, db.ptr, n)
    capi.synchronize()
    mine = (bins_of >= 3072) & (bins_of < 4096)
    q, qb = values[mine][:40000], bins_of[mine][:40000].astype(np.int64) - 3072
    g1 = ix.probe(q)
    assert np.all((g1[np.arange(q.size), qb >> 6] >> (qb & 63).astype(np.uint64)) & np.uint64(1))
    # k-mers that were only inserted into OTHER shards' bins are (almost surely) absent here
    other = values[~mine][:40000]
    assert ix.probe(other).any(axis=1).mean() < 0.01
    ix.emplace_device(dv.ptr, db.ptr, n)
    capi.synchronize()
    assert np.array_equal(ix.probe(q), g1)
    # a second copy filled in two halves in the other order holds the same bits on the probed rows
    iy = capi.Index.create_ibf(bins, m, h, shard_rank=3, n_shards=8)
    half = n // 2
    d2v, d2b = capi.DeviceBuffer.from_numpy(values[half:]), capi.DeviceBuffer.from_numpy(bins_of[half:])
    iy.emplace_device(d2v.ptr, d2b.ptr, n - half)
    d1v, d1b = capi.DeviceBuffer.from_numpy(values[:half]), capi.DeviceBuffer.from_numpy(bins_of[:half])
    iy.emplace_device(d1v.ptr, d1b.ptr, half)
    capi.synchronize()
    fresh = splitmix64(13, 30000) >> np.uint64(24)
    assert np.array_equal(iy.probe(q), g1) and np.array_equal(iy.probe(fresh), ix.probe(fresh))
    ix.free(); iy.free()


def test_more_than_two_to_the_32_rows(capi, oracle):
    """bin_size >= 2^32 takes its own probe kernel (probe_bigrows_kernel) and the 64-bit branch of
    fastrange; 40 bins x (2^32 + 12345) rows = one word per row, 34 GB on the device and in the oracle.
    Values are inserted on both sides with their own emplace and the probes must agree bit for bit."""
    bins, m, h = 40, (1 << 32) + 12345, 3
    ox = oracle.Index.ibf(bins, m, h, dna=False, k=4)
    ix = capi.Index.create_ibf(bins, m, h)
    vals = splitmix64(11, 4000)
    bins_of = (splitmix64(12, 4000) % np.uint64(bins)).astype(np.uint32)
    for b in range(bins):
        ox.emplace(vals[bins_of == b], b)
    dv, db = capi.DeviceBuffer.from_numpy(vals), capi.DeviceBuffer.from_numpy(bins_of)
    ix.emplace_device(dv.ptr, db.ptr, vals.size)
    capi.synchronize()
    probes = np.concatenate([vals, splitmix64(13, 4000)])
    got, want = ix.probe(probes), ox.probe(probes)
    assert np.array_equal(got, want)
    assert all((int(got[i, 0]) >> int(bins_of[i])) & 1 for i in range(vals.size))  # no false negatives
    assert int(np.count_nonzero(got[vals.size:])) == 0  # 12000 bits in 2^32 rows: a random probe finds nothing
    ix.free()


def test_queries_beyond_eight_million_ops_keep_streaming(capi, monkeypatch):
    """Round 1 gave up on a query after 8 M mask operations (CompileLimits), where the reference would — slowly — answer.
    Now only the states held at one time are bounded and the ops stream to the device stage by stage.  Known-answer
    index: every bit of bins 0..63 set, bins 64..127 empty, so no state ever dies in word 0 and the candidate mask of
    any searchable motif is exactly word 0 = all ones, word 1 = 0.  Murphy alphabet, k = 6: a run of eight wildcards
    keeps 10^5 suffix states alive and costs 2 M ops per position when the states are enumerated (TETREX_DENSE=0)."""
    bins, m, k = 128, 257, 6
    words = np.zeros((m, 2), dtype=np.uint64)
    words[:, 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
    ix = capi.Index.upload_ibf(bins, m, 3, words.reshape(-1))
    motifs = ["LMKDEF........HKLMNP", "LMAEGLYN"]  # (the reduced-alphabet builder has no X{m,n} over a union: eight dots)
    for dense in ("1", "0"):
        monkeypatch.setenv("TETREX_DENSE", dense)
        got, status, stats = ix.query_masks(motifs, False, k, reduction=1)
        assert status == [0, 0], (dense, status)
        for g in got:
            assert int(g[0]) == 0xFFFFFFFFFFFFFFFF and int(g[1]) == 0
        if dense == "0":
            assert stats["ops"] > (8 << 20) and stats["stages"] > 1 and stats["dense_ops"] == 0
        else:
            assert stats["dense_ops"] > 0 and stats["ops"] < (1 << 20)
    ix.free()
