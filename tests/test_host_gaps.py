"""-a / -g on the CPU: the C++ host's k-graph augmentation (Gap nodes) and d-gram probing against the
oracle's restatement of augment() / gap_procedure / update_gapped / DGramIndex
(reference include/otf_collector.h:216-245,290-312,395-493, include/dGramIndex.h).  The device
session is the numpy simulator; the d-gram index is its auxiliary index."""
import numpy as np
import pytest

from helpers import SessionSimulator

AA = "ACDEFGHIKLMNPQRSTVWY"
PLANTED = "LMAEGKRDEWWYNLMAHHCDEFPQRSTVKLMN"

# Gap sets with more than two lengths lose all but the first and last one in the reference (a Split has
# two successor slots); which survive depends on hash-set order there, so parity uses <= 2 lengths.
GAPPED = ["LMAEG.{4}WWYN", "LMAEG.{5}WWYN", "LMAEGKRDEWWYN", "LMAE.{5}WWYN", "LMAEG.{3,4}WWYN", "LMA.{2}KRDE.{3}NLMA",
          "LMAEG[KR].{3}WWYN", "KRDE.{2}YNLM.{2}HCDE", "EGKR.{4,5}NLMA", "CDEF.{4}TVKL", "LMAEG.{4}WWYN.{2}AHHC",
          "G.{3}EWWY", "LMAEG(K|R).{2}EWWY", "AEG.{9}MAHH", "LMAEGKRDEWWYNLMAHH"]


@pytest.fixture(scope="module")
def host():
    from tetrex_amd import host as H
    H.lib()
    return H


@pytest.fixture(scope="module")
def library(oracle, host):
    rng = np.random.default_rng(0)
    bins = 96
    seqs = ["".join(rng.choice(list(AA), size=400)) for _ in range(bins)]
    for b in (5, 40, 77):
        at = int(rng.integers(20, 300))
        seqs[b] = seqs[b][:at] + PLANTED + seqs[b][at + len(PLANTED):]
    seqs[41] = seqs[41][:50] + PLANTED[:9] + "A" + PLANTED[10:] + seqs[41][50 + len(PLANTED):]  # a near miss
    ox = oracle.Index.ibf(bins, oracle.compute_bitcount(400, 0.05), 3, dna=False, k=4)
    for b, s in enumerate(seqs):
        ox.emplace(oracle.decompose(s, 4, dna=False), b)
    lo, hi = 1, 10
    codes = [oracle.dgram_codes(s, lo, hi) for s in seqs]
    dg = oracle.Index.ibf(bins, oracle.compute_bitcount(max(len(c) for c in codes), 0.05), 3, dna=False, k=4)
    for b, c in enumerate(codes):
        dg.emplace(c, b)
    return dict(bins=bins, seqs=seqs, ox=ox, dg=dg, lo=lo, hi=hi, codes=codes)


def test_dgram_codes_match_the_oracle(host, oracle, library):
    for s in library["seqs"][:10] + ["ACD", "ACDEFGHIK", "ACDXFGHIKLMNPQRS", "", "ACDEFGHIKLMNPQRSTVWYacdefgh"]:
        for lo, hi in ((1, 10), (3, 21), (0, 2), (5, 5)):
            assert np.array_equal(host.dgram_values(s, lo, hi), oracle.dgram_codes(s, lo, hi))


@pytest.mark.parametrize("with_dgram", [False, True])
@pytest.mark.parametrize("per_query", [0, 3])
def test_augmented_queries_equal_the_oracle(host, oracle, library, with_dgram, per_query):
    ox, dg = library["ox"], library["dg"] if with_dgram else None
    gaps = dict(augment=1, dgram_loaded=int(with_dgram), min_gap=library["lo"] if with_dgram else 0,
                max_gap=library["hi"] if with_dgram else 0)
    sim = SessionSimulator(ox, len(GAPPED), dg)
    status, stats = host.run_staged(GAPPED, False, 4, 0, library["bins"], sim.stage, per_query, 0, gaps=gaps)
    assert not any(status)
    gapped = filtered = 0
    for i, q in enumerate(GAPPED):
        want, st = ox.query_aug(q, True, dg, gaps["min_gap"], gaps["max_gap"])
        if st["quirk_merges"]:
            continue
        assert np.array_equal(sim.result(i), want), q
        gapped += st["gap_nodes"] > 0
        filtered += st["dgram_probes"] > 0
        plain = ox.query(q)
        if not with_dgram:  # without the d-gram index a gap only relaxes the filter
            assert np.array_equal(plain & want, plain), q
    assert gapped >= 8
    assert (filtered >= 6) == with_dgram


def test_without_augment_gap_options_change_nothing(host, oracle, library):
    ox = library["ox"]
    sim = SessionSimulator(ox, len(GAPPED), library["dg"])
    host.run_staged(GAPPED, False, 4, 0, library["bins"], sim.stage, gaps=dict(augment=0, dgram_loaded=1, min_gap=1, max_gap=10))
    for i, q in enumerate(GAPPED):
        want, st = ox.query(q, with_stats=True)
        if st["quirk_merges"] == 0:
            assert np.array_equal(sim.result(i), want)


def test_the_dgram_filter_removes_a_bin_the_plain_gap_keeps(host, oracle, library):
    """LMAEG.{5}WWYN: the planted text has 4 residues between LMAEG and WWYN, so the plain query finds
    nothing; -a alone skips the gap region and keeps the planted bins as candidates; -a -g probes
    the (gap=5) d-gram and drops them again."""
    ox, dg = library["ox"], library["dg"]
    q = "LMAEG.{5}WWYN"
    bits = lambda m: [b for b in range(library["bins"]) if (int(m[b >> 6]) >> (b & 63)) & 1]
    assert len(set(bits(ox.query(q))) & {5, 40, 77}) <= 1
    sim = SessionSimulator(ox, 1)
    host.run_staged([q], False, 4, 0, library["bins"], sim.stage, gaps=dict(augment=1))
    assert {5, 40, 77} <= set(bits(sim.result(0)))
    sim = SessionSimulator(ox, 1, dg)
    host.run_staged([q], False, 4, 0, library["bins"], sim.stage, gaps=dict(augment=1, dgram_loaded=1, min_gap=1, max_gap=10))
    assert len(set(bits(sim.result(0))) & {5, 40, 77}) <= 1  # a Bloom false positive may keep one
