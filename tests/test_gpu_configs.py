"""Parity at the shapes of BASELINE.json configs[2..4] (configs[0] and [1] are covered in
test_gpu_query.py / test_gpu_probe.py / bench.py): the GPU path against the CPU oracle, bit-exact.
Sizes are scaled so the oracle finishes in seconds; row counts (Bloom filter sizes) shrink, bin
counts and tree shapes do not."""
import numpy as np
import pytest

from helpers import regular_hibf, splitmix64
from motifs import random_prosite_motifs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from tetrex_amd import capi as c
    c.init(0)
    return c


def _popcount(m):
    return int(np.unpackbits(np.ascontiguousarray(m).view(np.uint8)).sum())


def test_config3_1024_bin_hibf_with_a_batch_of_1k_motifs(capi, oracle):
    """configs[2]: 1024-bin HIBF (k=4 peptides), 1k PROSITE-style motifs in one streamed batch.  Every bin holds the
    4-mers of its own random sequence and every motif is a window of some bin's sequence with residue classes, wildcards
    and x(m,n) gaps worked in (the bench batch's mix), so a motif has a home bin it must be found in and most masks are
    neither empty nor full — a descent that returned zeros, or ones, would fail here."""
    from tetrex_amd import host
    rng = np.random.default_rng(3)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    seqs = [aa[rng.integers(0, 20, size=1503)].tobytes() for _ in range(1024)]
    ox, descs, values = regular_hibf(oracle, 1024, 64, 1500, lambda b: host.record_values_array(seqs[b], 4, dna=False), h=3)
    ix = capi.Index.upload_hibf(1024, descs)
    assert ix.info.n_ibf == 65 and ix.info.is_hibf
    motifs, home = [], []
    for _ in range(1000):
        b = int(rng.integers(0, 1024))
        n = int(rng.integers(6, 13))
        at = int(rng.integers(0, 1503 - n))
        w = [chr(c) for c in seqs[b][at:at + n]]
        for p in range(1, n - 1):  # both ends stay literal so that trimming does not eat the motif
            r = rng.random()
            if r < 0.10:
                w[p] = "."
            elif r < 0.40:
                others = rng.choice([c for c in "ACDEFGHIKLMNPQRSTVWY" if c != w[p]], size=int(rng.integers(1, 5)), replace=False)
                w[p] = "[" + "".join(sorted(set([w[p]]) | set(others))) + "]"
            elif r < 0.45:
                lo = int(rng.integers(0, 2))
                w[p] = ".{%d,%d}" % (lo, lo + int(rng.integers(1, 3)))  # the one residue that was here fits the gap
        motifs.append("".join(w))
        home.append(b)
    got, status, stats = ix.query_masks(motifs, False, 4)
    assert not any(status)
    checked = informative = skipped = 0
    for i, rx in enumerate(motifs):
        assert (int(got[i, home[i] >> 6]) >> (home[i] & 63)) & 1, rx  # no false negatives: the home bin is a candidate
        want, quirks = ox.expected_mask(rx)
        skipped += quirks > 0  # compared all the same, against the result under well-defined merges
        assert np.array_equal(got[i], want), rx
        checked += 1
        informative += 0 < _popcount(want) < 1024
    assert skipped < 100, skipped          # the reference's result is well defined for > 90 % of the batch
    assert checked > 900 and informative > 850, (checked, informative)
    assert sum("." in m for m in motifs) > 300
    ix.free()


def test_config4_8192_bin_dna_ibf_in_8_column_shards(capi, oracle):
    """configs[3]: 8192-bin DNA IBF sharded 8 ways by bin columns (1024 bins per shard); the shards run
    one after the other on this single GPU and their masks are concatenated like the all-gather does."""
    bins, m, h, k = 8192, 20011, 3, 8
    ox = oracle.Index.ibf(bins, m, h, dna=True, k=k)
    n = 1 << 21
    vals = splitmix64(4, n) >> np.uint64(48)
    ox.emplace_pairs(vals, (splitmix64(5, n) % np.uint64(bins)).astype(np.uint32))
    rng = np.random.default_rng(9)

    def motif():
        parts = []
        for _ in range(int(rng.integers(9, 16))):
            r = rng.random()
            parts.append("[" + "".join(rng.choice(list("ACGT"), size=2, replace=False)) + "]" if r < 0.2 else
                         (str(rng.choice(list("ACGT"))) + "?" if r < 0.25 else str(rng.choice(list("ACGT")))))
        return "".join(parts)
    motifs = [motif() for _ in range(300)]
    full = np.zeros((len(motifs), bins // 64), dtype=np.uint64)
    words = ox.words()
    for r in range(8):
        ix = capi.Index.upload_ibf(bins, m, h, words, shard_rank=r, n_shards=8)
        assert ix.shard_words == 16 and int(ix.info.shard_word0) == 16 * r
        got, status, _ = ix.query_masks(motifs, True, k)
        assert not any(status)
        full[:, 16 * r:16 * (r + 1)] = got
        ix.free()
    checked = 0
    for i, rx in enumerate(motifs):
        want, ost = ox.query(rx, with_stats=True)
        if ost["quirk_merges"] == 0:
            assert np.array_equal(full[i], want), rx
            checked += 1
    assert checked > 250


def test_config5_65536_bin_hibf_reduced_alphabet_k5(capi, oracle):
    """configs[4]: 65536-user-bin peptide HIBF, Murphy alphabet, k=5, two levels of 256-wide IBFs."""
    rng = np.random.default_rng(5)
    codes = np.arange(10, dtype=np.uint64)

    def vals(b):
        sym = rng.integers(0, 10, size=(40, 5))
        return (codes[sym] << (np.uint64(5) * np.arange(4, -1, -1, dtype=np.uint64))).sum(axis=1).astype(np.uint64)
    ox, descs, values = regular_hibf(oracle, 65536, 256, 40, vals, h=2, k=5, reduction=1)
    ix = capi.Index.upload_hibf(65536, descs)
    assert ix.info.n_ibf == 257 and ix.info.mask_words == 1024
    # (1) level-by-level descent of a k-mer batch
    kmers = np.concatenate([np.array([values[b][0] for b in range(0, 65536, 97)], dtype=np.uint64), vals(0)[:200]])
    got = ix.probe(kmers)
    want = ox.probe(kmers)
    assert np.array_equal(got, want)
    for j, b in enumerate(range(0, 65536, 97)):
        assert (int(got[j, b >> 6]) >> (b & 63)) & 1
    # (2) whole queries in the reduced alphabet: spell motifs from inserted k-mers
    letters = "ABCFGHIKPS"  # Murphy representatives by code 0..9

    def spell(v):
        return "".join(letters[(int(v) >> s) & 31] for s in (20, 15, 10, 5, 0))
    motifs = [spell(values[b][1]) + spell(values[b][2])[-1] for b in range(5, 65536, 4099)] + ["LMA(E|Q)GLYN", "IIABG", "KRDEGL"]
    res, status, _ = ix.query_masks(motifs, False, 5, reduction=1)
    assert not any(status)
    for i, rx in enumerate(motifs):
        assert np.array_equal(res[i], ox.query(rx)), rx
    ix.free()
