"""Dense DP steps on the GPU (csrc/txq_exec.hip dense_kernel; host/compiler.cpp densify / dense_step): whole
queries through the C++ host (libtetrex_query.so -> txq session), with the thresholds forced so low that nearly
every list becomes a dense block, against the CPU oracle's collect() (reference
include/otf_collector.h:341-393 over bulk_contains).  Bit-exact, on several mask widths, shards and both molecules."""
import numpy as np
import pytest

from motifs import PEPTIDE_QUERIES, DNA_QUERIES, random_prosite_motifs

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _lists_saturate(monkeypatch):
    """These tests are about the dense machinery, on small (sparse) indexes: tell the expansion that lists saturate instead
    of letting it find out that they do not (tests of that protocol set TETREX_DENSE_EVIDENCE themselves)."""
    monkeypatch.setenv("TETREX_DENSE_EVIDENCE", "dense")


@pytest.fixture(scope="module")
def capi():
    from tetrex_amd import capi as c
    c.init(0)
    return c


def _oracle_index(oracle, bins, m, h, k, dna, per_bin, seed, reduction=0):
    ox = oracle.Index.ibf(bins, m, h, dna=dna, k=k, reduction=reduction)
    rng = np.random.default_rng(seed)
    bits = (2 if dna else 5) * k
    for b in range(bins):
        ox.emplace(rng.integers(0, 1 << min(bits, 62), size=per_bin, dtype=np.uint64), b)
    return ox


def _wants(ox, queries):
    out = []
    for q in queries:
        try:
            out.append(ox.expected_mask(q)[0])
        except Exception:
            out.append(False)  # the reference path cannot search it either
    return out


TRACKED = [0]  # tracked queries of the last _check


def _check(capi, ox, queries, dna, k, reduction=0, shards=(1,), wants=None, per_query=0):
    sh = ox.shape()
    wants = wants if wants is not None else _wants(ox, queries)
    checked, dense_ops = 0, 0
    TRACKED[0] = 0
    for R in shards:
        for r in range(R):
            ix = capi.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words(), shard_rank=r, n_shards=R)
            lo, nw = int(ix.info.shard_word0), ix.shard_words
            got, status, stats = ix.query_masks(queries, dna, k, reduction, per_query)
            dense_ops += stats["dense_ops"]
            TRACKED[0] += stats["tracked_queries"]
            for q, g, w, st in zip(queries, got, wants, status):
                if w is False:
                    assert st != 0, q
                    continue
                assert st == 0, q
                if w is not None:
                    assert np.array_equal(g, w[lo:lo + nw]), (q, R, r)
                    checked += 1
            ix.free()
    return checked, dense_ops


@pytest.mark.parametrize("knobs", [("1", "0"), ("2", "2"), ("24", "8"), (None, None)], ids=["everything", "2/2", "24/8", "defaults"])
def test_peptide_queries_dense_vs_oracle(capi, oracle, monkeypatch, knobs):
    if knobs[0]:
        monkeypatch.setenv("TETREX_DENSE_MIN", knobs[0])
        monkeypatch.setenv("TETREX_DENSE_SPARSE_BELOW", knobs[1])
    ox = _oracle_index(oracle, bins=1024, m=4099, h=3, k=4, dna=False, per_bin=1500, seed=1)
    qs = PEPTIDE_QUERIES + random_prosite_motifs(60, 7, wildcard=0.1, ranges=0.05)
    wants = _wants(ox, qs)
    checked, dense_ops = _check(capi, ox, qs, False, 4, shards=(1, 4), wants=wants)
    assert checked > 300 and dense_ops > 50
    # and the same batch with dense steps switched off gives the same masks (A/B of the two paths)
    monkeypatch.setenv("TETREX_DENSE", "0")
    checked2, none = _check(capi, ox, qs, False, 4, wants=wants)
    assert checked2 * 5 == checked and none == 0


def test_dense_on_odd_and_wide_masks(capi, oracle, monkeypatch):
    """Mask widths that take the other lane layouts of dense_kernel: 1 word, an odd number of words (8-byte lanes),
    130 words (more 16-byte chunks than lanes per suffix)."""
    monkeypatch.setenv("TETREX_DENSE_MIN", "2")
    monkeypatch.setenv("TETREX_DENSE_SPARSE_BELOW", "2")
    qs = ["LMK.{1,3}A[DE]..GK", "WKL..[LIVM]D.[FY]", "LMKA.C.E.GH", "KRK[RK]{2,3}.DE", "CLM.{2,4}C...[LIVMFYWC]", "LMA(E|Q)GLYN"]
    for bins, per_bin in ((40, 700), (64 * 3 - 5, 700), (64 * 7, 500), (8300, 120)):
        ox = _oracle_index(oracle, bins=bins, m=2053, h=3, k=4, dna=False, per_bin=per_bin, seed=bins)
        checked, dense_ops = _check(capi, ox, qs, False, 4, shards=(1, 3) if bins > 200 else (1,))
        assert checked >= 5 and dense_ops > 20, bins


def test_dna_and_reduced_alphabets_dense(capi, oracle, monkeypatch):
    monkeypatch.setenv("TETREX_DENSE_MIN", "1")
    monkeypatch.setenv("TETREX_DENSE_SPARSE_BELOW", "0")
    for bins, m, k, per_bin, h in ((70, 257, 3, 8, 3), (300, 4099, 5, 300, 2), (128, 8191, 7, 900, 4)):
        ox = _oracle_index(oracle, bins=bins, m=m, h=h, k=k, dna=True, per_bin=per_bin, seed=bins)
        qs = DNA_QUERIES + ["ACG..T.GA", "A.{2,4}CGT.A", "AC[GT]..[AC]CGT"]
        checked, dense_ops = _check(capi, ox, qs, True, k)
        assert checked > 8 and dense_ops > 10
    for red in (1, 2):
        ox = _oracle_index(oracle, bins=256, m=8191, h=2, k=5, dna=False, per_bin=1500, seed=red, reduction=red)
        qs = ["LMA(E|Q)GLYN", "LMAEGLYNK", "W[LIVM]D.FYLK", "LMAE(GL|YN)K.DE", "KRDEG..NLMA"]
        checked, dense_ops = _check(capi, ox, qs, False, 5, red)
        assert checked >= 3 and dense_ops > 5


@pytest.mark.parametrize("tracked", [False, True], ids=["blocks", "tracked"])
def test_steps_through_the_table_of_all_kmer_masks(capi, oracle, monkeypatch, tracked):
    """Index::kmer_table (csrc/txq_exec.hip ensure_kmer_table, TableRows): dense steps read bulk_contains of every packed
    k-mer value from a table built once per index instead of gathering rows — DNA (canonical k-mers: the steps look the
    canonical value up), reduced alphabets (4-bit codes), peptide k = 4 on column shards, full and tracked blocks.  Masks
    against the oracle, table on (built by a session of any size here: TXQ_KMER_TABLE_MIN=1) and off."""
    monkeypatch.setenv("TETREX_DENSE_MIN", "1")
    monkeypatch.setenv("TETREX_DENSE_SPARSE_BELOW", "0")
    monkeypatch.setenv("TXQ_KMER_TABLE_MIN", "1")
    if tracked:
        monkeypatch.setenv("TETREX_DENSE_TRACKED", "1")
    cases = []
    for bins, m, k, per_bin, h in ((70, 257, 3, 8, 3), (300, 4099, 5, 300, 2), (128, 8191, 7, 900, 4)):
        cases.append((_oracle_index(oracle, bins=bins, m=m, h=h, k=k, dna=True, per_bin=per_bin, seed=bins),
                      DNA_QUERIES + ["ACG..T.GA", "A.{2,4}CGT.A", "AC[GT]..[AC]CGT"], True, k, 0, (1,)))
    for red in (1, 2):
        cases.append((_oracle_index(oracle, bins=256, m=8191, h=2, k=5, dna=False, per_bin=1500, seed=red, reduction=red),
                      ["LMA(E|Q)GLYN", "LMAEGLYNK", "W[LIVM]D.FYLK", "LMAE(GL|YN)K.DE", "KRDEG..NLMA"], False, 5, red, (1,)))
    cases.append((_oracle_index(oracle, bins=700, m=8191, h=3, k=4, dna=False, per_bin=2500, seed=9),
                  ["LMK.{1,3}A[DE]..GK", "WKL..[LIVM]D.[FY]", "LMKA.C.E.GH", "CLM.{2,4}C...[LIVMFYWC]", "A.CD", "K[RK]DE"], False, 4, 0, (1, 2)))
    for ox, qs, dna, k, red, shards in cases:
        wants = _wants(ox, qs)
        for table_mb in ("512", "0"):
            monkeypatch.setenv("TXQ_KMER_TABLE_MB", table_mb)
            checked, dense_ops = _check(capi, ox, qs, dna, k, red, shards=shards, wants=wants)
            assert checked >= 3 and dense_ops > 5, (dna, k, red, table_mb)
            if tracked:
                assert TRACKED[0] > 0


def test_dense_blocks_across_stages_and_recycling(capi, oracle, monkeypatch):
    """Tiny per-query stage budgets: blocks are written in one stage and read in the next; regions grow while they
    hold live blocks (device-to-device copy) and blocks are recycled."""
    monkeypatch.setenv("TETREX_DENSE_MIN", "3")
    monkeypatch.setenv("TETREX_DENSE_SPARSE_BELOW", "2")
    ox = _oracle_index(oracle, bins=512, m=2053, h=3, k=4, dna=False, per_bin=800, seed=5)
    qs = ["LMK.{1,3}A[DE]..GK", "WKL..[LIVM]D.[FY]", "LMKA.C.E.GH", "CLM.{2,4}C...[LIVMFYWC]", "LMK.{0,2}C.{0,2}D.{0,2}EK"]
    for per_query in (3, 40):
        checked, dense_ops = _check(capi, ox, qs, False, 4, per_query=per_query)
        assert checked == len(qs) and dense_ops > 30


@pytest.mark.parametrize("task_ops", [None, "300"])
def test_waves_of_queries_on_a_session_that_is_never_waited_for(capi, oracle, monkeypatch, task_ops):
    """Small waves (TETREX_WAVE_OPS): many stages none of which asks a question, so the host submits stage n+1 (other
    staging set, upload stream) while the kernels of stage n run; with a small task budget the queries also CONTINUE
    across those stages (regions grow and move, blocks live from one stage to the next).  Masks equal the oracle's."""
    monkeypatch.setenv("TETREX_WAVE_OPS", "500")
    monkeypatch.setenv("TETREX_WAVE_GROWTH", "0")  # (waves of 500 ops throughout; by default a wave is at least as large as all before it)
    if task_ops:
        monkeypatch.setenv("TETREX_TASK_OPS", task_ops)
    ox = _oracle_index(oracle, bins=1000, m=30011, h=3, k=4, dna=False, per_bin=3000, seed=21)
    qs = random_prosite_motifs(120, 21, wildcard=0.12, ranges=0.08)
    wants = _wants(ox, qs)
    sh = ox.shape()
    ix = capi.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words())
    for rep in range(2):  # the second session adopts the first one's staging sets and stream
        got, status, stats = ix.query_masks(qs, False, 4, 0, 0)
        for q, g, w, st in zip(qs, got, wants, status):  # (masks first: how a batch is cut into stages depends on host timing)
            if w is False:
                assert st != 0, q
            else:
                assert st == 0 and np.array_equal(g, w), q
        assert stats["stages"] >= 2 and stats["dense_ops"] > 50
    # the count of stages under a schedule that does not depend on timing: one expansion thread, nothing beside the stage
    monkeypatch.setenv("TETREX_THREADS", "1")
    monkeypatch.setenv("TETREX_NO_OVERLAP", "1")
    got1, status1, stats1 = ix.query_masks(qs, False, 4, 0, 0)
    assert list(status1) == list(status) and np.array_equal(got1, got)
    assert stats1["stages"] >= 4 and stats1["dense_ops"] > 50
    ix.free()


def test_saturated_motifs_cost_few_host_ops(capi, oracle):
    """Defaults: the wildcard-rich motifs of the bench batch.  Masks equal the oracle's and the host emits a small
    fraction of the ops it needs without dense steps."""
    import os
    ox = _oracle_index(oracle, bins=1024, m=4099, h=3, k=4, dna=False, per_bin=1500, seed=9)
    qs = [q for q in random_prosite_motifs(300, 6) if "." in q][:40]
    sh = ox.shape()
    ix = capi.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words())
    got, status, stats = ix.query_masks(qs, False, 4)
    os.environ["TETREX_DENSE"] = "0"
    try:
        got0, status0, stats0 = ix.query_masks(qs, False, 4)
    finally:
        del os.environ["TETREX_DENSE"]
    ix.free()
    assert status == status0 and np.array_equal(got, got0)
    assert stats["dense_ops"] > 0 and stats0["dense_ops"] == 0 and stats["ops"] * 10 < stats0["ops"]
    compared = 0
    for q, g, st in zip(qs[:12], got, status):
        m, ost = ox.query(q, with_stats=True)
        if st == 0 and not ost["quirk_merges"]:
            assert np.array_equal(g, m), q
            compared += 1
    assert compared >= 6


@pytest.mark.parametrize("tree", ["irregular-3-levels", "16x64", "40x128", "16x64-generic", "40x128-generic", "12x64-mixed", "24x128-mixed",
                                  "40x128-2shards", "6x256", "16x64-general", "40x128-general", "80x64", "70x128-2shards", "16x64-bylane", "6x256-bylane", "16x64-2shards", "3x64"])
def test_dense_steps_on_hibf_indexes(capi, oracle, monkeypatch, tree):
    """Dense steps on an HIBF session.  Regular two-level trees (the layout `tetrex index` writes) run a step fused like
    a flat IBF: root bit and child rows gathered per lane (txq_exec.hip TreeRows / TreeRowsByLane for roots of at most 64
    merged bins / InterleavedRows for small trees of uniform children; one-word and wider children, children of
    different row and hash counts, column shards).  Other trees — and regular ones with TXQ_DENSE_TREE=0 — write the
    predecessor k-mers of a step out, descend them as one batch (whatever descent kernel the tree takes) and combine
    (dense_hibf_*).  Masks against the oracle's collect() over membership_for, thresholds forced low."""
    from helpers import random_hibf, regular_hibf
    rng = np.random.default_rng(17)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    from tetrex_amd import host
    if tree == "irregular-3-levels":
        ox, descs, values = random_hibf(oracle, 21, user_bins=300, levels=3, n_values=60)
        ub = 300
        seqs = None

        def spell(v):
            return "".join("ACDEFGHIKLMNPQRSTVWY"[(int(v) >> sh) & 31] for sh in (15, 10, 5, 0))
        planted = [spell(v) for b in range(0, 300, 23) for v in values[b][:3] if all(((int(v) >> sh) & 31) < 20 for sh in (15, 10, 5, 0))]
    else:
        planted = []
        shape, _, variant = tree.partition("-")
        children, per_child = (int(x) for x in shape.split("x"))
        ub = children * per_child
        if variant == "mixed":
            ub -= 5  # the last child is not full
        if variant == "generic":
            monkeypatch.setenv("TXQ_DENSE_TREE", "0")
        if variant == "bylane":  # small uniform trees have their children interleaved; this is what they would run otherwise
            monkeypatch.setenv("TXQ_DENSE_TREE", "2")
        if variant == "general":  # the variant for large trees (every lane gathers the root words of its own child), forced on a small one
            monkeypatch.setenv("TXQ_DENSE_TREE", "1")
        seqs = [aa[rng.integers(0, 20, size=203)].tobytes() for _ in range(ub)]
        ox, descs, values = regular_hibf(oracle, ub, children, 200, lambda b: host.record_values_array(seqs[b], 4, dna=False), h=2, mixed=variant == "mixed")
    qs = ["LMK.{1,3}A[DE]..GK", "WKL..[LIVM]D.[FY]", "LMKA.C.E.GH", "KRK[RK]{2,3}.DE", "CLM.{2,4}C...[LIVMFYWC]", "LMA(E|Q)GLYN", "A.CD", "K[RK]DE"]
    qs += planted[:12] + [p[0] + "." + p[2:] for p in planted[:6]] + [p[:2] + "[" + "".join(sorted(set(p[2] + "AK"))) + "]" + p[3] for p in planted[6:12]]
    if seqs is not None:  # windows of the bins' own sequences with wildcards: non-trivial masks
        for b in range(0, ub, max(1, ub // 12)):
            w = [chr(c) for c in seqs[b][50:60]]
            w[3] = "."
            w[5] = "[" + "".join(sorted(set([w[5], "A", "K"]))) + "]"
            w[7] = ".{0,2}"
            qs.append("".join(w))
    n_shards = 2 if tree.endswith("2shards") else 1
    wants = [ox.query(q, with_stats=True) for q in qs]
    for rank in range(n_shards):
        ix = capi.Index.upload_hibf(ub, descs, shard_rank=rank, n_shards=n_shards)
        lo, nw = int(ix.info.shard_word0), ix.shard_words
        # the tree's own step kernels first (the index's table of all k-mers' masks off), then, last, the steps through that table
        for knobs in (("2", "2"), ("1", "0"), None, ("2", "2", "table")):
            monkeypatch.setenv("TXQ_KMER_TABLE_MB", "512" if knobs and len(knobs) == 3 else "0")
            monkeypatch.setenv("TXQ_KMER_TABLE_MIN", "1")
            if knobs:
                monkeypatch.setenv("TETREX_DENSE_MIN", knobs[0])
                monkeypatch.setenv("TETREX_DENSE_SPARSE_BELOW", knobs[1])
            else:
                monkeypatch.delenv("TETREX_DENSE_MIN")
                monkeypatch.delenv("TETREX_DENSE_SPARSE_BELOW")
            got, status, stats = ix.query_masks(qs, False, 4)
            assert stats["dense_ops"] > 0
            hits = 0
            for q, g, st, (want, ost) in zip(qs, got, status, wants):
                assert st == 0, q
                if not ost["quirk_merges"]:
                    assert np.array_equal(g, want[lo:lo + nw]), (q, knobs, rank)
                    hits += int(want.any())
            assert hits >= 3
        ix.free()


def test_large_blocks_k5_base_alphabet_and_k6_murphy(capi, oracle, monkeypatch):
    """Blocks of 21^4 = 194 481 slots (k = 5, Base alphabet) and 10^5 slots (k = 6, Murphy): the dense slot ids reach
    beyond 2^20, ZERO / REDUCE tiles number in the hundreds, and the regions of a query take tens of MB."""
    monkeypatch.setenv("TETREX_DENSE_MIN", "2")
    monkeypatch.setenv("TETREX_DENSE_SPARSE_BELOW", "2")
    ox = _oracle_index(oracle, bins=130, m=8191, h=3, k=5, dna=False, per_bin=1200, seed=55)
    qs = ["LMKDE..[KR]G.HK", "WKLMN[LIVM].D[FY]..K", "LMKDA.C.E.GH", "CLMKD.{1,2}C..[LIVMFYWC]K"]
    checked, dense_ops = _check(capi, ox, qs, False, 5)
    assert checked == len(qs) and dense_ops > 10
    ox = _oracle_index(oracle, bins=96, m=8191, h=2, k=6, dna=False, per_bin=1500, seed=56, reduction=1)
    qs = ["LMKDEF..[KR]G.HKL", "WKLMNP[LIVM].D[FY]..KD"]
    checked, dense_ops = _check(capi, ox, qs, False, 6, 1)
    assert checked >= 1 and dense_ops > 4


def test_dense_regions_are_recycled_when_block_memory_is_short(capi, oracle, monkeypatch):
    """TETREX_DENSE_POOL_MB=200 (about 170 blocks of a 1024-bin index) for 80 wildcard motifs: queries are admitted in waves,
    finished ones hand their blocks back and the device gives their dense regions to the next wave (txq_exec.hip
    grow_slot_regions).  Same masks as with the default pool, and as the oracle's."""
    ox = _oracle_index(oracle, bins=1024, m=4099, h=3, k=4, dna=False, per_bin=1500, seed=31)
    qs = [q for q in random_prosite_motifs(400, 41, wildcard=0.15, ranges=0.08) if "." in q][:80]
    sh = ox.shape()
    ix = capi.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words())
    ref, status0, stats0 = ix.query_masks(qs, False, 4)
    monkeypatch.setenv("TETREX_DENSE_POOL_MB", "200")
    monkeypatch.setenv("TXQ_TRACE", "1")
    got, status, stats = ix.query_masks(qs, False, 4)
    ix.free()
    assert status == status0 and np.array_equal(got, ref)
    # (a query that needs more blocks than the pool has left falls back to enumerated states for that list: more ops, same masks)
    assert stats["stages"] > stats0["stages"] and stats["dense_ops"] > 0 and stats["ops"] < 10 * stats0["ops"] + 10000
    compared = 0
    for q, g, st in zip(qs[:25], got, status):
        m, ost = ox.query(q, with_stats=True)
        if st == 0 and not ost["quirk_merges"]:
            assert np.array_equal(g, m), q
            compared += 1
    assert compared >= 12


@pytest.mark.parametrize("kind", ["saturated", "sparse"])
def test_a_fresh_index_is_asked_how_states_fare_on_it(capi, oracle, monkeypatch, kind):
    """TETREX_DENSE_EVIDENCE unset, index tag 0 (a fresh index in the product): the first wildcard query pauses before its first block and reads the
    fill of the probed states' masks from the device's answers (1 + floor(log2(bits))).  Saturated index: dense steps, tag 1;
    sparse index: a higher bar for blocks (QueryExpansion::shape_limit), tag 2 — or 3 where the states are down to a handful of bins (tracked blocks); the next batch starts from the tag (no pause).  Masks equal the oracle's."""
    monkeypatch.delenv("TETREX_DENSE_EVIDENCE")
    bins, k = 1000, 3
    if kind == "saturated":
        ox = oracle.Index.ibf(bins, 4099, 2, dna=False, k=k)
        every = np.arange(1 << 15, dtype=np.uint64)
        for b in range(bins):
            if b % 5:
                ox.emplace(every, b)
    else:
        ox = _oracle_index(oracle, bins=bins, m=4099, h=2, k=k, dna=False, per_bin=300, seed=31)
    qs = ["LMKA..CDE.GH", "WKLA.{1,3}CDEF", "ACDEF...GHIKL", "LMKACDE", "LMK[AC]..[DE]F.HK"]
    wants = _wants(ox, qs)
    sh = ox.shape()
    ix = capi.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words())
    assert ix.tag == 0
    stages = []
    for rep in range(2):
        got, status, stats = ix.query_masks(qs, False, k, 0, 0)
        for q, g, w, st in zip(qs, got, wants, status):
            assert st == 0 and np.array_equal(g, w), (q, rep)
        assert stats["dense_ops"] > 0 or kind == "sparse"
        assert (ix.tag & 3 == 1) if kind == "saturated" else (ix.tag & 3 in (2, 3))  # (3: states do not just thin out, they die out: tracked blocks)
        stages.append(stats["stages"])
        asked = stats["dense_ops"]
    assert stages[1] <= stages[0]
    ix.tag = 2 if kind == "saturated" else 1  # told the opposite, the expansion believes it (another bar for blocks): the same masks
    got, status, stats = ix.query_masks(qs, False, k, 0, 0)
    for q, g, w, st in zip(qs, got, wants, status):
        assert st == 0 and np.array_equal(g, w), q
    assert ix.tag & 3 == (2 if kind == "saturated" else 1)  # what an index is known for is not revised by a run that did not ask
    # where states thin out the bar for a block is four times higher (narrow masks: some lists still clear it)
    if kind == "saturated":
        assert stats["dense_ops"] <= asked
    ix.free()


@pytest.mark.parametrize("index", ["flat", "hibf-16x64"])
def test_every_way_of_running_a_batch_gives_the_same_masks(capi, oracle, monkeypatch, index):
    """One batch of 160 PROSITE-style motifs through the combinations of the knobs that choose HOW it runs — enumerated or
    dense, one wave or many, waves queued or side by side on two streams, unit ops in their own launch or riding in the
    dense one, tiny per-query budgets (programs that continue across stages), the tree variants of the dense step and of
    the plain probe — must give, bit for bit, the masks of the plainest way (enumerated, one wave), which the oracle checks."""
    from helpers import regular_hibf
    from tetrex_amd import host
    rng = np.random.default_rng(77)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    qs = random_prosite_motifs(160, 41, wildcard=0.12, ranges=0.06)
    if index == "flat":
        ox = _oracle_index(oracle, bins=1000, m=30011, h=3, k=4, dna=False, per_bin=3000, seed=21)
        sh = ox.shape()
        ix = capi.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words())
    else:
        seqs = [aa[rng.integers(0, 20, size=400)].tobytes() for _ in range(1024)]
        ox, descs, _ = regular_hibf(oracle, 1024, 16, 400, lambda b: host.record_values_array(seqs[b], 4, dna=False), h=2)
        ix = capi.Index.upload_hibf(1024, descs)
    plain = {"TETREX_DENSE": "0", "TETREX_WAVE_OPS": "0"}
    ways = [plain,
            {"TETREX_DENSE_EVIDENCE": "dense"},
            {"TETREX_DENSE_EVIDENCE": "dense", "TETREX_WAVE_OPS": "3000"},
            {"TETREX_DENSE_EVIDENCE": "dense", "TETREX_WAVE_OPS": "3000", "TXQ_ONE_STREAM": "1"},
            {"TETREX_DENSE_EVIDENCE": "dense", "TETREX_WAVE_OPS": "800", "TETREX_TASK_OPS": "200", "TXQ_FUSE_UNITS": "0"},
            {"TETREX_DENSE_EVIDENCE": "sparse", "TETREX_WAVE_OPS": "3000"},
            {"TETREX_DENSE_EVIDENCE": "ask", "TETREX_WAVE_OPS": "5000"},
            {"TETREX_DENSE_EVIDENCE": "dense", "TETREX_DENSE_MIN": "4", "TETREX_DENSE_SPARSE_BELOW": "2", "TETREX_WAVE_OPS": "2000"},
            {"TETREX_DENSE_EVIDENCE": "dense", "TXQ_KMER_TABLE_MB": "0"},  # rows gathered (the other ways read the table of all k-mers' masks)
            {"TETREX_DENSE_EVIDENCE": "dense", "TXQ_KMER_TABLE_MB": "0", "TETREX_WAVE_OPS": "3000", "TXQ_DENSE_UNROLL": "5"}]
    if index != "flat":
        ways += [{"TETREX_DENSE_EVIDENCE": "dense", "TXQ_DENSE_TREE": t, "TETREX_WAVE_OPS": "3000"} for t in ("0", "1", "2")]
        ways += [{"TETREX_DENSE_EVIDENCE": "dense", "TXQ_HIBF_INTERLEAVE_PROBE": "0", "TETREX_WAVE_OPS": "3000"}]
    knobs = sorted({k for w in ways for k in w})
    want = None
    for way in ways:
        for k in knobs:
            if k in way:
                monkeypatch.setenv(k, way[k])
            else:
                monkeypatch.delenv(k, raising=False)
        ix.tag = 0
        got, status, stats = ix.query_masks(qs, False, 4, 0, 0)
        if want is None:
            want, want_status = got, status
            checked = 0
            for q, g, st in zip(qs[:40], got, status):  # the plainest way against the oracle
                try:
                    w, _ = ox.expected_mask(q)
                except Exception:
                    assert st != 0
                    continue
                assert st == 0 and np.array_equal(g, w), q
                checked += 1
            assert checked >= 30
        else:
            assert list(status) == list(want_status), way
            assert np.array_equal(got, want), way
            if way.get("TETREX_DENSE_EVIDENCE") == "dense":
                assert stats["dense_ops"] > 0, way
    ix.free()


def test_random_wave_sizes_budgets_and_block_pools(capi, oracle, monkeypatch):
    """tools/stress_waves.py in small: one batch under random wave sizes, per-query budgets and block pools — stages that
    run beside each other on two streams, programs that continue across stages, regions that are given back and reused —
    always the masks of the plainest run (one wave, one stream), which is checked against the oracle."""
    ox = _oracle_index(oracle, bins=1000, m=30011, h=3, k=4, dna=False, per_bin=3000, seed=33)
    sh = ox.shape()
    ix = capi.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words())
    qs = random_prosite_motifs(200, 57, wildcard=0.12, ranges=0.06)
    monkeypatch.setenv("TETREX_WAVE_OPS", "0")
    monkeypatch.setenv("TXQ_ONE_STREAM", "1")
    want, status, _ = ix.query_masks(qs, False, 4, 0, 0)
    checked = 0
    for q, g, st in zip(qs[:30], want, status):
        try:
            w, _ = ox.expected_mask(q)
        except Exception:
            assert st != 0
            continue
        assert st == 0 and np.array_equal(g, w), q
        checked += 1
    assert checked >= 20
    monkeypatch.delenv("TXQ_ONE_STREAM")
    rng = np.random.default_rng(5)
    for it in range(16):
        monkeypatch.setenv("TETREX_WAVE_OPS", str(int(rng.integers(200, 4000))))  # (the batch is some 15 000 ops)
        monkeypatch.setenv("TETREX_WAVE_GROWTH", str(int(rng.choice([0, 50, 100]))))
        if rng.random() < 0.5:
            monkeypatch.setenv("TETREX_TASK_OPS", str(int(rng.integers(100, 3000))))
        else:
            monkeypatch.delenv("TETREX_TASK_OPS", raising=False)
        monkeypatch.setenv("TETREX_DENSE_POOL_MB", str(int(rng.choice([100, 1000, 49152]))))
        monkeypatch.setenv("TETREX_THREADS", str([1, 3, 16][it % 3]))  # (forced cuts of the expansion over host threads)
        got, st, stats = ix.query_masks(qs, False, 4, 0, 0)
        assert list(st) == list(status) and np.array_equal(got, want), it
        assert stats["dense_ops"] > 0
    ix.free()


# ---- tracked (sparse) blocks: steps pushed from the live lists (csrc/txq_exec.hip sparse_kernel) ----------------------

@pytest.fixture(params=["compacted", "lane-groups"])
def step_kernel(request, monkeypatch):
    """Pushed steps on a flat index run compacted (sparse_step_kernel: items queued in LDS, one per lane) or, with
    TXQ_SPARSE_STEPS=0, in sparse_kernel with a lane group per entry like the trees' — both must give the oracle's masks."""
    if request.param == "lane-groups":
        monkeypatch.setenv("TXQ_SPARSE_STEPS", "0")
    return request.param


def test_tracked_blocks_vs_oracle(capi, oracle, monkeypatch, step_kernel):
    """TETREX_DENSE_TRACKED=1: every query keeps its blocks with live lists; STEP / REDUCE / ZERO / FILL follow the lists
    (sparse_plan_kernel + sparse_kernel).  Same masks as the oracle and as the run without dense blocks, on a flat index, on
    column shards, with nearly every list a block."""
    monkeypatch.setenv("TETREX_DENSE_TRACKED", "1")
    monkeypatch.setenv("TETREX_DENSE_MIN", "2")
    monkeypatch.setenv("TETREX_DENSE_SPARSE_BELOW", "2")
    ox = _oracle_index(oracle, bins=1024, m=4099, h=3, k=4, dna=False, per_bin=1500, seed=1)
    qs = PEPTIDE_QUERIES + random_prosite_motifs(60, 7, wildcard=0.1, ranges=0.05)
    wants = _wants(ox, qs)
    checked, dense_ops = _check(capi, ox, qs, False, 4, shards=(1, 4), wants=wants)
    assert checked > 300 and dense_ops > 50 and TRACKED[0] > 100
    monkeypatch.setenv("TETREX_DENSE_MIN", "32")
    monkeypatch.setenv("TETREX_DENSE_SPARSE_BELOW", "16")
    checked, dense_ops = _check(capi, ox, qs, False, 4, wants=wants)
    assert checked > 60 and dense_ops > 20


def test_tracked_blocks_on_odd_and_wide_masks_dna_and_reduced_alphabets(capi, oracle, monkeypatch, step_kernel):
    monkeypatch.setenv("TETREX_DENSE_TRACKED", "1")
    monkeypatch.setenv("TETREX_DENSE_MIN", "2")
    monkeypatch.setenv("TETREX_DENSE_SPARSE_BELOW", "2")
    qs = ["LMK.{1,3}A[DE]..GK", "WKL..[LIVM]D.[FY]", "LMKA.C.E.GH", "KRK[RK]{2,3}.DE", "CLM.{2,4}C...[LIVMFYWC]", "LMA(E|Q)GLYN"]
    for bins, per_bin in ((40, 700), (64 * 3 - 5, 700), (64 * 7, 500), (8300, 120)):
        ox = _oracle_index(oracle, bins=bins, m=2053, h=3, k=4, dna=False, per_bin=per_bin, seed=bins)
        checked, dense_ops = _check(capi, ox, qs, False, 4, shards=(1, 3) if bins > 200 else (1,))
        assert checked >= 5 and dense_ops > 20, bins
    for bins, m, k, per_bin, h in ((70, 257, 3, 8, 3), (300, 4099, 5, 300, 2), (128, 8191, 7, 900, 4)):
        ox = _oracle_index(oracle, bins=bins, m=m, h=h, k=k, dna=True, per_bin=per_bin, seed=bins)
        dq = DNA_QUERIES + ["ACG..T.GA", "A.{2,4}CGT.A", "AC[GT]..[AC]CGT"]
        checked, dense_ops = _check(capi, ox, dq, True, k)
        assert checked > 8 and dense_ops > 10
    for red in (1, 2):
        ox = _oracle_index(oracle, bins=256, m=8191, h=2, k=5, dna=False, per_bin=1500, seed=red, reduction=red)
        rq = ["LMA(E|Q)GLYN", "LMAEGLYNK", "W[LIVM]D.FYLK", "LMAE(GL|YN)K.DE", "KRDEG..NLMA"]
        checked, dense_ops = _check(capi, ox, rq, False, 5, red)
        assert checked >= 3 and dense_ops > 5 and TRACKED[0] >= 2


def test_tracked_blocks_across_stages_and_recycling(capi, oracle, monkeypatch, step_kernel):
    """Tiny stage budgets with tracked blocks: a block's list is written in one stage and read in the next; finished
    programs hand their blocks to later ones (cleared before they are used again)."""
    monkeypatch.setenv("TETREX_DENSE_TRACKED", "1")
    monkeypatch.setenv("TETREX_DENSE_MIN", "3")
    monkeypatch.setenv("TETREX_DENSE_SPARSE_BELOW", "2")
    monkeypatch.setenv("TETREX_WAVE_OPS", "300")
    ox = _oracle_index(oracle, bins=512, m=2053, h=3, k=4, dna=False, per_bin=800, seed=5)
    qs = ["LMK.{1,3}A[DE]..GK", "WKL..[LIVM]D.[FY]", "LMKA.C.E.GH", "CLM.{2,4}C...[LIVMFYWC]", "LMK.{0,2}C.{0,2}D.{0,2}EK"] * 6
    for per_query in (3, 40, 0):
        checked, dense_ops = _check(capi, ox, qs, False, 4, per_query=per_query)
        assert checked == len(qs) and dense_ops > 30 and TRACKED[0] >= 20


def test_k6_wildcard_motifs_run_as_tracked_blocks(capi, oracle, monkeypatch, step_kernel):
    """The reference's default k (include/arg_parse.h:12) on the Base alphabet: 21^5 suffixes per block, of which a sparse
    index keeps a few alive.  With the run told that states thin out (what it learns by asking) wildcard motifs become
    tracked blocks — a handful of ops instead of one per state and residue — and give the oracle's masks, identical to the
    run without dense blocks."""
    monkeypatch.setenv("TETREX_DENSE_EVIDENCE", "thin")
    k = 6
    ox = oracle.Index.ibf(1024, 20011, 3, dna=False, k=k)
    rng = np.random.default_rng(66)
    from tetrex_amd import host as H
    seqs = []
    for b in range(1024):  # bins of random residues with a few motif instances planted, so that paths survive
        s = "".join("ACDEFGHIKLMNPQRSTVWY"[i] for i in rng.integers(0, 20, size=700))
        if b % 97 == 0:
            s = s[:100] + "LMKACDEGHW" + s[110:300] + "CAAKCLLLMAAAAAAAAH" + s[318:]
        seqs.append(s)
        ox.emplace(H.record_values_array(s, k, dna=False), b)
    # (motifs whose states the CPU oracle can enumerate in a moment: a literal k-mer or two in front of the wildcards)
    qs = ["LMK..DEGH", "LMKA..E.H", "KCLL.{2,4}A..[AH]", "L[MK]K.C.[DE]G", "KAC.{1,3}GHW", "LMK.{2,4}GH", "LMKAC.{0,3}W", "CAAK..L.MA.A.....H",
          "KCLL.{1,3}A.{2,4}A.H"]
    wants = _wants(ox, qs)
    sh = ox.shape()
    ix = capi.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words())
    got, status, stats = ix.query_masks(qs, False, k, 0, 0)
    assert stats["dense_ops"] > 10 and stats["ops"] < 200000 and stats["tracked_queries"] >= 4
    monkeypatch.setenv("TETREX_DENSE", "0")
    plain, status0, stats0 = ix.query_masks(qs, False, k, 0, 0)
    for q, g, p0, w, st in zip(qs, got, plain, wants, status):
        assert st == 0 and w is not False, q
        assert np.array_equal(g, p0), q
        assert np.array_equal(g, w), q
    assert any(int(np.bitwise_or.reduce(g)) for g in got)
    ix.free()


def test_k6_motif_batch_tracked_blocks_equal_enumerated_states(capi, oracle, monkeypatch, step_kernel):
    """A batch of PROSITE-style motifs with wildcards and x(m,n) gaps at k = 6 on 1024 bins of random sequences (the shape
    of tests/perf_cli_swissprot_shape.py, smaller): nothing is told about the index — the run asks, learns that states thin
    out, and keeps wildcard lists as tracked blocks.  Masks identical to the run without dense blocks (every state
    enumerated and pruned by the host), and to the oracle where it answers quickly."""
    monkeypatch.setenv("TETREX_DENSE_EVIDENCE", "ask")
    k = 6
    from tetrex_amd import host as H
    ox = oracle.Index.ibf(1024, 60013, 3, dna=False, k=k)
    rng = np.random.default_rng(67)
    for b in range(1024):
        s = "".join("ACDEFGHIKLMNPQRSTVWY"[i] for i in rng.integers(0, 20, size=3000))
        ox.emplace(H.record_values_array(s, k, dna=False), b)
    qs = random_prosite_motifs(120, 3, wildcard=0.08, ranges=0.04, min_len=8, max_len=14)
    sh = ox.shape()
    ix = capi.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words())
    got, status, stats = ix.query_masks(qs, False, k, 0, 0)
    assert stats["tracked_queries"] >= 10 and stats["dense_ops"] > 50
    monkeypatch.setenv("TETREX_DENSE", "0")
    plain, status0, stats0 = ix.query_masks(qs, False, k, 0, 0)
    assert list(status) == list(status0)
    assert np.array_equal(got, plain)
    assert stats["ops"] * 20 < stats0["ops"]
    compared = 0
    for q, g, st in zip(qs, got, status):
        if "." in q[:10] or st:
            continue
        assert np.array_equal(g, ox.expected_mask(q)[0]), q
        compared += 1
    assert compared > 30
    ix.free()


@pytest.mark.parametrize("tree", ["16x64", "8x256-mixed", "4x64-bylane"])
def test_tracked_blocks_on_regular_hibfs(capi, oracle, monkeypatch, tree):
    """Tracked programs on regular two-level trees: the pushed steps take their rows from the tree (sparse_kernel with the
    InterleavedRows / TreeRows policies, the shapes `tetrex index` writes: the README scenario runs exactly this).  Masks =
    the oracle's collect() over membership_for."""
    from helpers import regular_hibf
    from tetrex_amd import host
    monkeypatch.setenv("TETREX_DENSE_TRACKED", "1")
    monkeypatch.setenv("TETREX_DENSE_MIN", "2")
    monkeypatch.setenv("TETREX_DENSE_SPARSE_BELOW", "2")
    rng = np.random.default_rng(19)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    shape, _, variant = tree.partition("-")
    children, per_child = (int(x) for x in shape.split("x"))
    ub = children * per_child - (5 if variant == "mixed" else 0)
    if variant == "bylane":
        monkeypatch.setenv("TXQ_DENSE_TREE", "1")  # no interleaved children: TreeRows
    seqs = [aa[rng.integers(0, 20, size=203)].tobytes() for _ in range(ub)]
    ox, descs, values = regular_hibf(oracle, ub, children, 200, lambda b: host.record_values_array(seqs[b], 4, dna=False), h=2, mixed=variant == "mixed")
    qs = ["LMK.{1,3}A[DE]..GK", "WKL..[LIVM]D.[FY]", "LMKA.C.E.GH", "KRK[RK]{2,3}.DE", "LMA(E|Q)GLYN", "A.CD"]
    for b in range(0, ub, max(1, ub // 10)):
        w = [chr(c) for c in seqs[b][50:60]]
        w[3] = "."
        w[5] = "[" + "".join(sorted(set([w[5], "A", "K"]))) + "]"
        w[7] = ".{0,2}"
        qs.append("".join(w))
    wants = [ox.query(q, with_stats=True) for q in qs]
    ix = capi.Index.upload_hibf(ub, descs)
    for table_mb in ("0", "512"):  # the tree's rows; then the same steps through the index's table of all k-mers' masks
        monkeypatch.setenv("TXQ_KMER_TABLE_MB", table_mb)
        monkeypatch.setenv("TXQ_KMER_TABLE_MIN", "1")
        got, status, stats = ix.query_masks(qs, False, 4)
        assert stats["tracked_queries"] >= 8 and stats["dense_ops"] > 20
        hits = 0
        for q, g, st, (want, ost) in zip(qs, got, status, wants):
            assert st == 0, q
            if not ost["quirk_merges"]:
                assert np.array_equal(g, want), (q, table_mb)
                hits += int(want.any())
    assert hits >= 5
    ix.free()


def test_emplace_after_a_dense_batch_drops_the_table_of_kmer_masks(capi, oracle, monkeypatch):
    """the index caches bulk_contains of every packed k-mer value (Index::kmer_table, built by the first dense
    batch); txq_emplace_device changes the bits under it.  Half the values, a wildcard batch (table built), the other half, the
    batch again: both runs must equal the oracle on the bits of their time."""
    monkeypatch.setenv("TETREX_DENSE_EVIDENCE", "dense")
    monkeypatch.setenv("TXQ_KMER_TABLE_MIN", "1")
    bins, m, h, k = 200, 4001, 3, 4
    rng = np.random.default_rng(17)
    values = rng.integers(0, 1 << 20, size=60000, dtype=np.uint64)
    bins_of = rng.integers(0, bins, size=values.size, dtype=np.uint32)
    qs = ["L..A", "A.C.E", "W..[LIVM]D", "K[RK].DE", "C..C.", "[ST].[RK].A", "LM.{1,3}A[DE]", "A.C.E.GH"] * 3
    ox = oracle.Index.ibf(bins, m, h, dna=False, k=k)
    ix = capi.Index.create_ibf(bins, m, h)
    half = values.size // 2
    for lo, hi in ((0, half), (half, values.size)):
        ox.emplace_pairs(values[lo:hi], bins_of[lo:hi])
        dv, db = capi.DeviceBuffer.from_numpy(values[lo:hi]), capi.DeviceBuffer.from_numpy(bins_of[lo:hi])
        ix.emplace_device(dv.ptr, db.ptr, hi - lo)
        capi.synchronize()
        got, status, stats = ix.query_masks(qs, False, k)
        assert stats["dense_ops"] > 0
        informative = 0
        for q, g, st in zip(qs, got, status):
            assert st == 0
            want, _ = ox.expected_mask(q)
            assert np.array_equal(g, want), (q, lo)
            informative += int(want.any())
        assert informative >= 9
        assert np.array_equal(ix.download_words_rows(m), ox.words())
    ix.free()


def test_blocks_are_pooled_with_the_index_from_batch_to_batch(capi, oracle, monkeypatch):
    """All blocks of a session go back into a pool kept with the index (csrc/txq_internal.hpp SessionCache::blocks); the next
    batch takes its blocks from there, and a block that a tracked program left behind — all zero outside its live list — is
    taken over by a tracked program WITHOUT being cleared: the ZERO that creates the block clears what is listed.  Batches of
    different motifs, tracked and untracked in turn, on one index: every batch must give the oracle's masks."""
    monkeypatch.setenv("TETREX_DENSE_MIN", "2")
    monkeypatch.setenv("TETREX_DENSE_SPARSE_BELOW", "2")
    ox = _oracle_index(oracle, bins=1024, m=4099, h=3, k=4, dna=False, per_bin=1500, seed=21)
    sh = ox.shape()
    ix = capi.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words())
    batches = [random_prosite_motifs(40, seed, wildcard=0.12, ranges=0.06) for seed in (31, 32, 33)]
    wants = [_wants(ox, b) for b in batches]
    order = [(0, "1"), (1, "1"), (0, "1"), (2, None), (1, "1"), (0, None), (2, "1"), (2, "1")]
    for which, tracked in order:
        if tracked:
            monkeypatch.setenv("TETREX_DENSE_TRACKED", tracked)
        else:
            monkeypatch.delenv("TETREX_DENSE_TRACKED", raising=False)
        got, status, stats = ix.query_masks(batches[which], False, 4)
        assert stats["dense_ops"] > 20
        assert (stats["tracked_queries"] > 10) == bool(tracked)
        for q, g, w, st in zip(batches[which], got, wants[which], status):
            if w is False:
                assert st != 0, q
            elif w is not None:
                assert st == 0 and np.array_equal(g, w), (q, which, tracked)
    ix.free()
