"""Every environment switch of libtxq.so (include/txq.h lists them) chooses between code paths that give the SAME
results: each one is set here in turn and a small workload — plain probes of a flat IBF and of three HIBF shapes, and
whole queries with dense steps — still has to give the CPU oracle's masks.  (The switches that deliberately computed
wrong masks, for timing experiments, are not in the product build any more: tests/test_capi_symbols.py.)"""
import numpy as np
import pytest

from helpers import regular_hibf

pytestmark = pytest.mark.gpu

KNOBS = [
    {}, {"TXQ_TRACE_SYNC": "1"}, {"TXQ_DENSE_TREE": "0"}, {"TXQ_DENSE_TREE": "1"}, {"TXQ_DENSE_TREE": "2"},
    {"TXQ_DENSE_UNROLL": "2"}, {"TXQ_DENSE_UNROLL": "6"}, {"TXQ_DENSE_SLICES": "1"}, {"TXQ_DENSE_SLICES": "4"},
    {"TXQ_DENSE_TILE_ROUNDS": "1"}, {"TXQ_DENSE_TILE_ROUNDS": "5"}, {"TXQ_FUSE_UNITS": "0"}, {"TXQ_ONE_STREAM": "1"},
    # (five queries: no table of all k-mers' masks by default — TXQ_KMER_TABLE_MIN=1 builds it; the 1024-bin table is 128 MB)
    {"TXQ_KMER_TABLE_MIN": "1"}, {"TXQ_KMER_TABLE_MIN": "1", "TXQ_DENSE_UNROLL": "5"}, {"TXQ_KMER_TABLE_MIN": "1", "TXQ_KMER_TABLE_MB": "64"},
    {"TXQ_KMER_TABLE_MIN": "1", "TXQ_KMER_TABLE_MB": "0"}, {"TXQ_KMER_TABLE_MIN": "1", "TXQ_DENSE_SLICES": "1", "TXQ_FUSE_UNITS": "0"},
    {"TXQ_HIBF_INTERLEAVE": "0"}, {"TXQ_HIBF_INTERLEAVE_PROBE": "0"}, {"TXQ_HIBF_LEVELS": "1"}, {"TXQ_HIBF_STATIONARY": "0"},
    {"TXQ_HIBF_SMALL": "0"}, {"TXQ_HIBF_LANE_HASH": "1"}, {"TXQ_HIBF_STEPS_PER_GROUP": "1"}, {"TXQ_HIBF_TILE": "256"},
    {"TXQ_HIBF_UNROLL": "2"}, {"TXQ_HIBF_UNROLL": "4"}, {"TXQ_HIBF_STORE_KIND": "1"}, {"TXQ_HIBF_STORE_KIND": "2"}, {"TXQ_HIBF_STORE_KIND": "3"},
    {"TXQ_HIBF_STORE": "48"},  # (the removed experiment switch: must change nothing)
    # pushed steps of tracked blocks (TETREX_DENSE_TRACKED=1 is the host's switch for them): by units, in rounds, and the units' knobs
    {"TETREX_DENSE_TRACKED": "1"}, {"TETREX_DENSE_TRACKED": "1", "TXQ_SPARSE_STEPS": "0"}, {"TETREX_DENSE_TRACKED": "1", "TXQ_SPARSE_UNROLL": "2"},
    {"TXQ_FINAL_PINNED": "0"},
    {"TETREX_DENSE_TRACKED": "1", "TXQ_SPARSE_UNITS": "64"}, {"TETREX_DENSE_TRACKED": "1", "TXQ_SPARSE_UNITS": "1536", "TXQ_KMER_TABLE_MIN": "1"},
    {"TXQ_HIBF_WAVES": "512"}, {"TXQ_PROBE_BLOCKS_PER_CU": "2"}, {"TXQ_PROBE_UNROLL": "1"}, {"TXQ_PROBE_UNROLL": "4"}, {"TXQ_PROBE_NT": "1"},
]


@pytest.fixture(scope="module")
def capi():
    from tetrex_amd import capi as c
    c.init(0)
    return c


@pytest.fixture(scope="module")
def workloads(oracle):
    rng = np.random.default_rng(77)
    flat = oracle.Index.ibf(1024, 8191, 3, dna=False, k=4)
    for b in range(1024):
        flat.emplace(rng.integers(0, 1 << 20, size=400, dtype=np.uint64), b)
    kmers = rng.integers(0, 1 << 20, size=5000, dtype=np.uint64)
    trees = []
    for user_bins, children in ((1024, 16), (4096, 32), (300, 5)):
        ox, descs, values = regular_hibf(oracle, user_bins, children, 60, lambda b: rng.integers(0, 1 << 20, size=60, dtype=np.uint64), h=2)
        trees.append((user_bins, ox, descs, np.concatenate([kmers[:1500]] + [v[:2] for v in values[:200]])))
    queries = ["LMK.{1,3}A[DE]..GK", "WKL..[LIVM]D.[FY]", "LMKA.C.E.GH", "LMA(E|Q)GLYN", "KRK[RK]{2,3}.DE"]
    want = {"flat_probe": flat.probe(kmers), "flat_queries": [flat.expected_mask(q)[0] for q in queries],
            "tree_probe": [ox.probe(km) for _, ox, _, km in trees], "tree_queries": [[ox.expected_mask(q)[0] for q in queries] for _, ox, _, _ in trees]}
    return flat, kmers, trees, queries, want


@pytest.mark.parametrize("knob", KNOBS, ids=lambda k: ",".join("%s=%s" % kv for kv in k.items()) or "defaults")
def test_masks_do_not_depend_on_a_knob(capi, workloads, monkeypatch, knob):
    flat, kmers, trees, queries, want = workloads
    for name, value in knob.items():
        monkeypatch.setenv(name, value)
    monkeypatch.setenv("TETREX_DENSE_EVIDENCE", "dense")
    monkeypatch.setenv("TETREX_DENSE_MIN", "4")
    sh = flat.shape()
    ix = capi.Index.upload_ibf(flat.bins, sh["bin_size"], sh["hash_funs"], flat.words())
    assert np.array_equal(ix.probe(kmers), want["flat_probe"])
    got, status, stats = ix.query_masks(queries, False, 4)
    assert all(s == 0 for s in status) and stats["dense_ops"] > 0
    for g, w in zip(got, want["flat_queries"]):
        assert np.array_equal(g, w)
    ix.free()
    for (user_bins, ox, descs, km), wp, wq in zip(trees, want["tree_probe"], want["tree_queries"]):
        ix = capi.Index.upload_hibf(user_bins, descs)
        assert np.array_equal(ix.probe(km), wp), user_bins
        got, status, stats = ix.query_masks(queries, False, 4)
        assert all(s == 0 for s in status)
        for g, w in zip(got, wq):
            assert np.array_equal(g, w), user_bins
        ix.free()
