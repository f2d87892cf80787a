"""Staged execution on the GPU: the C++ host expands the frontier piecewise and the device session
(txq_session_*) keeps the slot masks in HBM, answering the dead-state feedback.  Final masks must
equal the oracle's collect() for every stage budget, on flat IBFs, column shards and HIBFs."""
import numpy as np
import pytest

from motifs import PEPTIDE_QUERIES, DNA_QUERIES, random_prosite_motifs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from tetrex_amd import capi as c
    c.init(0)
    return c


@pytest.fixture(scope="module")
def host():
    from tetrex_amd import host as H
    return H


def _oracle_index(oracle, bins, m, h, k, dna, per_bin, seed):
    ox = oracle.Index.ibf(bins, m, h, dna=dna, k=k)
    rng = np.random.default_rng(seed)
    bits = (2 if dna else 5) * k
    for b in range(bins):
        ox.emplace(rng.integers(0, 1 << min(bits, 62), size=per_bin, dtype=np.uint64), b)
    return ox


def _staged(capi, host, ix, ox, queries, dna, k, per_query, per_stage=0):
    sess = ix.session(len(queries))
    status, stats = host.run_staged(queries, dna, k, 0, ox.bins, lambda blob, qp, qs: sess.stage(blob, qp, qs), per_query, per_stage)
    got = sess.end()
    lo, nw = int(ix.info.shard_word0), ix.shard_words
    checked = 0
    for i, q in enumerate(queries):
        try:
            want, ost = ox.query(q, with_stats=True)
        except Exception:
            assert status[i] != 0
            continue
        if ost["quirk_merges"] == 0:
            assert np.array_equal(got[i], want[lo:lo + nw]), (q, per_query)
            checked += 1
    return checked, stats


@pytest.mark.parametrize("per_query", [1, 16, 4096, 1 << 30])
def test_staged_session_matches_oracle(capi, host, oracle, per_query):
    ox = _oracle_index(oracle, bins=1024, m=4099, h=3, k=4, dna=False, per_bin=1500, seed=1)
    sh = ox.shape()
    ix = capi.Index.upload_ibf(1024, sh["bin_size"], 3, ox.words())
    qs = [q for q in PEPTIDE_QUERIES if "{2,4}C" not in q] + random_prosite_motifs(30, 8, wildcard=0.05, ranges=0.02)
    checked, stats = _staged(capi, host, ix, ox, qs, False, 4, per_query)
    assert checked >= len(qs) - 8
    if per_query == 1:
        assert stats["stages"] > 10
    ix.free()


def test_staged_feedback_prunes_on_a_sparse_index(capi, host, oracle):
    ox = _oracle_index(oracle, bins=300, m=60013, h=3, k=4, dna=False, per_bin=400, seed=2)
    ix = capi.Index.upload_ibf(300, 60013, 3, ox.words())
    qs = ["LMA.{2,4}E.{2}GLY", "W.{2}[LIVM]D[VFY][LIVM]{3}D.PPGT[GS]D", "C.{2,4}C.{3}[LIVMFYWC].{8}H.{3,5}H"]
    _, one = _staged(capi, host, ix, ox, qs, False, 4, 1 << 30)
    checked, st = _staged(capi, host, ix, ox, qs, False, 4, 512)
    assert checked >= 2 and st["pruned"] > 0 and st["ops"] < one["ops"] / 5
    ix.free()


def test_staged_on_column_shards_and_dna(capi, host, oracle):
    ox = _oracle_index(oracle, bins=300, m=4099, h=3, k=5, dna=True, per_bin=300, seed=5)
    for R in (1, 3):
        for r in range(R):
            ix = capi.Index.upload_ibf(300, 4099, 3, ox.words(), shard_rank=r, n_shards=R)
            checked, _ = _staged(capi, host, ix, ox, DNA_QUERIES, True, 5, per_query=8, per_stage=64)
            assert checked >= 10
            ix.free()


def test_staged_on_hibf(capi, host, oracle):
    from helpers import random_hibf
    ox, descs, values = random_hibf(oracle, 33, user_bins=300, levels=3, n_values=60)
    ix = capi.Index.upload_hibf(300, descs)
    qs = ["LMA(E|Q)GLYN", "A.CD", "K[RK]DE", "L.{1,2}KR", "ACDEF"]
    sess = ix.session(len(qs))
    status, stats = host.run_staged(qs, False, 4, 0, 300, lambda blob, qp, qs_: sess.stage(blob, qp, qs_), 32)
    got = sess.end()
    for i, q in enumerate(qs):
        assert np.array_equal(got[i], ox.query(q)), q
    ix.free()
