"""The C++ multi-shard product path (host/device_index.cpp ShardedStageExecutor / run_queries_sharded, exported as
txe_query_masks_sharded; `tetrex query --shards R`): ONE frontier expansion drives all column shards of an index, every
stage runs on all shards at the same time, pruning takes the OR of the shards' alive bits and the final masks are joined.
Here all shards share the one GPU of the box (txq_init with one device deals every shard onto it); on a multi-GPU node the
same code puts shard r on device r.  Full masks must equal the CPU oracle's (reference include/query.h:250-290 run_collection)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from motifs import PEPTIDE_QUERIES, DNA_QUERIES, random_prosite_motifs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from tetrex_amd import capi as c
    c.init(0)
    return c


def _oracle_index(oracle, bins, m, h, k, dna, per_bin, seed):
    ox = oracle.Index.ibf(bins, m, h, dna=dna, k=k)
    rng = np.random.default_rng(seed)
    bits = (2 if dna else 5) * k
    for b in range(bins):
        ox.emplace(rng.integers(0, 1 << min(bits, 62), size=per_bin, dtype=np.uint64), b)
    return ox


@pytest.mark.parametrize("R", [2, 3, 8])
def test_sharded_queries_equal_the_oracle(capi, oracle, R, monkeypatch):
    monkeypatch.setenv("TETREX_DENSE_EVIDENCE", "dense")  # a sparse index: the expansion would find out and enumerate; this test wants blocks on shards
    ox = _oracle_index(oracle, bins=1000, m=4099, h=3, k=4, dna=False, per_bin=1500, seed=5)
    sh = ox.shape()
    qs = PEPTIDE_QUERIES + random_prosite_motifs(40, 11, wildcard=0.1, ranges=0.05)
    shards = [capi.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words(), shard_rank=r, n_shards=R) for r in range(R)]
    full, status, stats = capi.query_masks_sharded(shards, qs, False, 4)
    assert full.shape == (len(qs), (ox.bins + 63) // 64)
    one = capi.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words())
    ref, status1, stats1 = one.query_masks(qs, False, 4)
    assert status == status1 and np.array_equal(full, ref)  # same masks as the unsharded index, query by query
    checked = 0
    for q, g, st in zip(qs, full, status):
        try:
            want, quirks = ox.expected_mask(q)
        except Exception:
            assert st != 0
            continue
        assert st == 0
        assert np.array_equal(g, want), q
        checked += 1
    assert checked > 50 and stats["dense_ops"] > 0
    for s in shards + [one]:
        s.free()


def test_sharded_feedback_prunes_with_the_or_of_all_shards(capi, oracle):
    """A sparse index and wildcard motifs: most states die, but a state that is dead in one shard and alive in another
    must survive.  Tiny stage budgets force many feedback rounds."""
    ox = _oracle_index(oracle, bins=640, m=60013, h=3, k=4, dna=False, per_bin=400, seed=2)
    sh = ox.shape()
    qs = ["LMA.{2,4}E.{2}GLY", "W.{2}[LIVM]D[VFY][LIVM]{3}D.PPGT[GS]D", "LMK.{1,3}A[DE]..GK", "LMAEGLYN"]
    os.environ["TETREX_DENSE"] = "0"  # enumerated states: the path that asks for feedback
    try:
        shards = [capi.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words(), shard_rank=r, n_shards=5) for r in range(5)]
        full, status, stats = capi.query_masks_sharded(shards, qs, False, 4, ops_per_query_per_stage=512)
    finally:
        del os.environ["TETREX_DENSE"]
    assert not any(status) and stats["pruned"] > 0 and stats["feedback_queries"] > 0
    for q, g in zip(qs, full):
        want, ost = ox.query(q, with_stats=True)
        if not ost["quirk_merges"]:
            assert np.array_equal(g, want), q
    for s in shards:
        s.free()


def test_dna_shards_with_an_empty_shard(capi, oracle):
    """More shards than mask words: some shards own no column at all and still take part."""
    ox = _oracle_index(oracle, bins=130, m=4099, h=3, k=5, dna=True, per_bin=300, seed=130)
    sh = ox.shape()
    shards = [capi.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words(), shard_rank=r, n_shards=4) for r in range(4)]
    assert sorted(s.shard_words for s in shards) == [0, 1, 1, 1]
    full, status, _ = capi.query_masks_sharded(shards, DNA_QUERIES, True, 5)
    for q, g, st in zip(DNA_QUERIES, full, status):
        try:
            want, ost = ox.query(q, with_stats=True)
        except Exception:
            assert st != 0
            continue
        if st == 0 and not ost["quirk_merges"]:
            assert np.array_equal(g, want), q
    for s in shards:
        s.free()


def test_cli_query_with_shards(tmp_path, oracle):
    """`tetrex query --shards 3` narrows to the same bins and prints the same verified matches as the unsharded run."""
    exe = os.path.join(ROOT, "bin", "tetrex")
    rng = np.random.default_rng(12)
    aa = list("ACDEFGHIKLMNPQRSTVWY")
    files = []
    for b in range(200):
        seq = "".join(rng.choice(aa, size=400))
        if b in (17, 150, 199):
            seq = seq[:100] + "LMAEGLYN" + seq[108:]
        p = tmp_path / ("bin%03d.fa" % b)
        p.write_text(">rec%d\n%s\n" % (b, seq))
        files.append(str(p))
    lst = tmp_path / "bins.lst"
    lst.write_text("\n".join(files) + "\n")
    subprocess.run([exe, "index", "-k", "4", "-i", str(tmp_path / "idx"), str(lst)], check=True, capture_output=True, timeout=300)
    outs = []
    for extra in ([], ["--shards", "3"], ["-D", "0", "--shards", "2"]):
        r = subprocess.run([exe, "query", "-v", *extra, str(tmp_path / "idx.ibf"), "LMA(E|Q)GLYN"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        narrowed = [ln for ln in r.stderr.splitlines() if ln.startswith("Narrowed Search")]
        outs.append((sorted(r.stdout.splitlines()), narrowed))
    assert outs[0] == outs[1] == outs[2]
    assert len(outs[0][0]) == 3 and all("LMAEGLYN" in ln for ln in outs[0][0])


def test_shards_of_a_fresh_index_are_asked_together(capi, oracle, monkeypatch):
    """Nothing known about the index: the expansion reads mask fills from the feedback answers, which the sharded executor
    adds up over the shards (1 + floor(log2(bits)) per shard -> of the sum).  Saturated index in 3 shards: blocks, the
    verdict lands in the first shard's tag, masks equal the oracle's."""
    monkeypatch.delenv("TETREX_DENSE_EVIDENCE", raising=False)
    bins, k = 640, 3
    ox = oracle.Index.ibf(bins, 4099, 2, dna=False, k=k)
    every = np.arange(1 << 15, dtype=np.uint64)
    for b in range(bins):
        if b % 3:
            ox.emplace(every, b)
    sh = ox.shape()
    qs = ["LMKA..CDE.GH", "WKLA.{1,3}CDEF", "ACDEF...GHIKL", "LMKACDE", "LMK[AC]..[DE]F.HK"]
    shards = [capi.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words(), shard_rank=r, n_shards=3) for r in range(3)]
    full, status, stats = capi.query_masks_sharded(shards, qs, False, k)
    for q, g, st in zip(qs, full, status):
        want, _ = ox.expected_mask(q)
        assert st == 0 and np.array_equal(g, want), q
    assert stats["dense_ops"] > 0 and shards[0].tag & 3 == 1
    for s in shards:
        s.free()


def test_shards_on_real_devices_when_the_box_has_several(oracle, monkeypatch):
    """txq_init(N, ids) with N real devices: shard r lives on device r, every session call binds the calling thread to the
    device of its index, the stage-submission threads (one per shard) talk to different GPUs at the same time.  Skipped on a
    one-GPU box (there every shard lands on the one device: the tests above)."""
    from tetrex_amd import capi as c
    n = c.device_count_safe()
    if n < 2:
        pytest.skip("one GPU: nothing to deal shards over")
    monkeypatch.setenv("TETREX_DENSE_EVIDENCE", "dense")
    c.init_devices(list(range(n)))
    try:
        ox = _oracle_index(oracle, bins=64 * 4 * n, m=4099, h=3, k=4, dna=False, per_bin=1200, seed=15)
        sh = ox.shape()
        qs = PEPTIDE_QUERIES + random_prosite_motifs(40, 12, wildcard=0.1, ranges=0.05)
        shards = [c.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words(), shard_rank=r, n_shards=n) for r in range(n)]
        assert sorted(int(s.info.device) for s in shards) == list(range(n))
        full, status, stats = c.query_masks_sharded(shards, qs, False, 4)
        checked = 0
        for q, g, st in zip(qs, full, status):
            try:
                want, quirks = ox.expected_mask(q)
            except Exception:
                assert st != 0
                continue
            assert st == 0 and np.array_equal(g, want), q
            checked += 1
        assert checked > 50
        # plain probes of every shard on its own device
        kmers = np.random.default_rng(3).integers(0, 1 << 20, size=4000, dtype=np.uint64)
        want = ox.probe(kmers)
        for s in shards:
            w0, nw = int(s.info.shard_word0), s.shard_words
            assert np.array_equal(s.probe(kmers), want[:, w0:w0 + nw])
        for s in shards:
            s.free()
    finally:
        c.init(0)
