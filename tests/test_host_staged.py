"""CPU tests of staged execution (host/compiler.cpp run_staged): the frontier is expanded piecewise
and dead states are pruned on feedback.  The device session is replaced by a numpy simulator over
oracle-probed masks; the final masks must equal the oracle's collect() for every stage budget."""
import os

import numpy as np
import pytest

from helpers import SessionSimulator
from motifs import PEPTIDE_QUERIES, DNA_QUERIES, random_prosite_motifs


@pytest.fixture(scope="module")
def host():
    from tetrex_amd import host as H
    H.lib()
    return H


def _index(oracle, bins, m, h, k, dna, per_bin, seed):
    ox = oracle.Index.ibf(bins, m, h, dna=dna, k=k)
    rng = np.random.default_rng(seed)
    bits = (2 if dna else 5) * k
    for b in range(bins):
        ox.emplace(rng.integers(0, 1 << min(bits, 62), size=per_bin, dtype=np.uint64), b)
    return ox


def _run(host, ox, queries, dna, k, per_query, per_stage=0):
    sim = SessionSimulator(ox, len(queries))
    status, stats = host.run_staged(queries, dna, k, 0, ox.bins, sim.stage, per_query, per_stage)
    checked = 0
    for i, q in enumerate(queries):
        try:
            want, quirks = ox.expected_mask(q)  # where the reference merges states of different length: the well-defined result
        except Exception:
            assert status[i] != 0
            continue
        assert status[i] == 0, q
        assert np.array_equal(sim.result(i), want), q
        checked += 1
    return checked, stats, sim


@pytest.mark.parametrize("per_query", [0, 1, 7, 64, 4096, 1 << 30])  # 0 = the product's default policy
def test_every_stage_budget_gives_the_oracle_masks(host, oracle, monkeypatch, per_query):
    if per_query == 1 << 30:
        monkeypatch.setenv("TETREX_WAVE_OPS", "0")  # (one stage: everybody begins at once, nobody pauses)
    ox = _index(oracle, bins=200, m=4099, h=3, k=4, dna=False, per_bin=1500, seed=1)
    qs = [q for q in PEPTIDE_QUERIES if "{2,4}C" not in q] + random_prosite_motifs(25, 3, wildcard=0.05, ranges=0.0)
    checked, stats, sim = _run(host, ox, qs, False, 4, per_query)
    assert checked >= len(qs) - 6
    if per_query == 1 << 30:
        assert stats["stages"] == 1 and stats["pruned"] == 0
    if per_query == 1:
        assert stats["stages"] > 10


def test_feedback_prunes_an_exploding_motif(host, oracle):
    """A wildcard-rich motif on a SPARSE index: without feedback the frontier is 20^(k-1) states per
    node; with feedback dead states are dropped and the op count collapses — same result mask."""
    ox = _index(oracle, bins=128, m=60013, h=3, k=4, dna=False, per_bin=400, seed=2)
    qs = ["LMA.{2,4}E.{2}GLY", "W.{2}[LIVM]D[VFY][LIVM]{3}D.PPGT[GS]D"]
    _, one_shot, _ = _run(host, ox, qs, False, 4, 1 << 30)
    _, staged, sim = _run(host, ox, qs, False, 4, 512)
    assert staged["pruned"] > 0 and staged["stages"] > 1
    assert staged["ops"] < one_shot["ops"] / 5
    for i, q in enumerate(qs):
        assert np.array_equal(sim.result(i), ox.query(q))


def test_stage_blob_budget_delays_queries_without_changing_results(host, oracle):
    ox = _index(oracle, bins=70, m=257, h=3, k=3, dna=True, per_bin=8, seed=3)
    checked, stats, _ = _run(host, ox, DNA_QUERIES, True, 3, per_query=4, per_stage=6)
    assert checked >= 10 and stats["stages"] >= 4


def test_one_bin_index_and_failed_queries(host, oracle):
    ox = oracle.Index.ibf(1, 64, 3, dna=False, k=4)
    sim = SessionSimulator(ox, 2)
    status, _ = host.run_staged(["LMAEGLYN", "ACDE"], False, 4, 0, 1, sim.stage)
    assert status == [0, 0] and int(sim.result(0)[0]) == 1 and int(sim.result(1)[0]) == 1
    ox = _index(oracle, bins=100, m=509, h=3, k=4, dna=False, per_bin=100, seed=4)
    sim = SessionSimulator(ox, 3)
    status, _ = host.run_staged(["LMAEG", "A{2,}", "LMAE"], False, 4, 0, 100, sim.stage)
    assert status[0] == 0 and status[1] != 0 and status[2] == 0
    assert np.array_equal(sim.result(0), ox.query("LMAEG")) and not sim.result(1).any()


def test_default_policy_on_sparse_and_dense_indexes_at_k5(host, oracle):
    """The product's default policy (adaptive budgets, expansion of confirmed states only where most
    states die, no merge table where a join's list has no mergeable pair) against the oracle at k = 5:
    a SPARSE index, where wildcard frontiers mostly die, and a DENSE one (every bit set: nothing dies,
    the lists after a wildcard run reach 20^4 states and beyond the merge-sample threshold)."""
    qs = ["LMA..E[DE]GLY", "WK.{1,2}[LIVM]D.F", "AC.DE.GH", "M[KR]..S[ST].L", "LMAEGLYN", "C.{2}C.H"]
    sparse = _index(oracle, bins=96, m=80021, h=3, k=5, dna=False, per_bin=300, seed=8)
    checked, stats, sim = _run(host, sparse, qs, False, 5, 0)
    assert checked == len(qs)
    wide = ["L...M...K", "A.{3}C.{2}DE", "W....[DE]K"]  # thousands of paths each: several stages, most states die
    checked, stats, sim = _run(host, sparse, wide, False, 5, 0)
    assert checked == len(wide) and stats["pruned"] > 0 and stats["stages"] >= 2
    dense = oracle.Index.ibf(64, 257, 3, dna=False, k=5)
    dense.set_words(np.full(257 * 1, np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64))
    checked, stats, sim = _run(host, dense, qs, False, 5, 0)
    assert checked == len(qs) and stats["pruned"] == 0
    for i in range(len(qs)):
        assert int(sim.result(i)[0]) == 0xFFFFFFFFFFFFFFFF


@pytest.mark.parametrize("per_query", [0, 3, 50, 1 << 30])
def test_node_for_node_graphs_and_fused_classes_give_the_same_masks(host, oracle, monkeypatch, per_query):
    """By default the expansion works on k-graphs whose unions of single residues are ONE node each (KGraph::kClass);
    TETREX_FUSE_CLASSES=0 keeps the reference's node per residue.  Both against the oracle, at budgets that stop a class
    between two of its residues (3 ops: after every residue).  In one stage (nothing pruned) the two probe exactly the same
    k-mers, and the fused graph never needs more ops (states that arrive at a class merge once, not once per residue)."""
    ox = _index(oracle, bins=150, m=20011, h=3, k=4, dna=False, per_bin=700, seed=21)
    qs = ["A.C.E", "L[LIVM].[DE]K", "W.{1,3}[KR]D", "(A|C|D)(E|F).G", "M[KR]+S.L", "A[CD]?E.[FGH]{2}K"] + \
        random_prosite_motifs(20, 13, wildcard=0.15, ranges=0.05)
    if per_query == 1 << 30:
        monkeypatch.setenv("TETREX_WAVE_OPS", "0")
    stats, probed = {}, {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("TETREX_FUSE_CLASSES", fuse)
        probed[fuse] = set()
        real_parse = host.parse_blob

        def spy(blob, into=probed[fuse]):
            kmers, progs = real_parse(blob)
            into.update(int(x) for x in kmers)
            return kmers, progs
        monkeypatch.setattr(host, "parse_blob", spy)
        checked, stats[fuse], _ = _run(host, ox, qs, False, 4, per_query)
        monkeypatch.setattr(host, "parse_blob", real_parse)
        assert checked >= len(qs) - 4
    if per_query == 1 << 30:
        assert probed["1"] == probed["0"] and len(probed["1"]) > 10000
        if "TETREX_MERGE_SAMPLE" not in os.environ:  # (test_host_variants.py: lists that do not merge at all — no such promise)
            assert stats["1"]["ops"] <= stats["0"]["ops"]


@pytest.mark.parametrize("dna", [False, True])
def test_plain_strings_expand_without_a_graph_to_the_same_ops(host, oracle, monkeypatch, dna):
    """A query that is a plain string of residues is expanded without a k-graph (QueryExpansion's literal constructor;
    TETREX_LITERAL_FAST=0: the general way): the same masks — the oracle's —, ops, k-mers and states, also for strings shorter
    than k (no probe: every bin) and of exactly k residues; an empty query fails in both."""
    if dna:
        ox = _index(oracle, bins=70, m=4099, h=2, k=6, dna=True, per_bin=300, seed=31)
        qs = ["ACGTACGTAC", "ACGTAC", "ACGT", "A", "GGGGGGGGGGGG", "ACGTNACGTAC", "TTTTTTT", ""]
        k = 6
    else:
        ox = _index(oracle, bins=130, m=4099, h=3, k=4, dna=False, per_bin=900, seed=32)
        qs = ["LMAEGLYN", "LMAE", "LM", "A", "ACDEFGHIKLMNPQRSTVWY", "LMAEXGLYN", "WWWWWW", "", "LMA.GLYN"]
        k = 4
    runs = {}
    for fast in ("1", "0"):
        monkeypatch.setenv("TETREX_LITERAL_FAST", fast)
        sim = SessionSimulator(ox, len(qs))
        status, stats = host.run_staged(qs, dna, k, 0, ox.bins, sim.stage)
        runs[fast] = (status, stats, [sim.result(i).copy() for i in range(len(qs))])
    (st1, s1, m1), (st0, s0, m0) = runs["1"], runs["0"]
    assert st1 == st0 and st1[qs.index("")] != 0
    for key in ("ops", "kmers", "states", "stages"):
        assert s1[key] == s0[key], (key, s1, s0)
    for i, q in enumerate(qs):
        assert np.array_equal(m1[i], m0[i]), q
        if st1[i] == 0:
            assert np.array_equal(m1[i], ox.query(q)), q


@pytest.mark.parametrize("wave_ops", ["0", "64", "2000"])
def test_waves_of_queries_give_the_same_masks(host, oracle, monkeypatch, wave_ops):
    """Queries begin in waves (TETREX_WAVE_OPS; the k-graph of a query is built when it begins), the later waves while
    the previous stage executes: more stages, the same masks, and a motif that fails to parse fails alone."""
    ox = _index(oracle, bins=130, m=4099, h=3, k=4, dna=False, per_bin=900, seed=11)
    qs = random_prosite_motifs(24, 5, wildcard=0.05, ranges=0.03)
    qs.insert(7, "AC(DE")  # syntax error in the middle of a wave
    monkeypatch.setenv("TETREX_WAVE_OPS", wave_ops)
    monkeypatch.setenv("TETREX_WAVE_GROWTH", "0")  # (waves of that size throughout; by default a wave is at least as large as all before it)
    checked, stats, sim = _run(host, ox, qs, False, 4, 0)
    assert checked >= len(qs) - 6
    if wave_ops == "0":
        test_waves_of_queries_give_the_same_masks.one_wave = stats
    elif wave_ops == "64":
        assert stats["stages"] >= 4
        one = getattr(test_waves_of_queries_give_the_same_masks, "one_wave", None)
        if one is not None:
            assert stats["stages"] > one["stages"]
