"""Sessions on GENERAL HIBFs — three and more levels, user bins next to merged bins, split bins, user bins in any order:
what seqan::hibf's layout produces (reference include/index_hibf.h:114-129,132-147) — work in LAYOUT ORDER
(csrc/txq_internal.hpp VChunk; csrc/txq_hibf.hip hibf_layout_level_kernel; csrc/txq_exec.hip PathRows): masks are the rows
of the tree's own technical bins, written segment by segment, dense steps gather a lane's bytes from one IBF behind its
ancestors' gates, and only the final masks are converted to user-bin order.  Every mask must equal the CPU oracle's
collect() over membership_for — and the masks of the same queries in user-bin order (TXQ_HIBF_LAYOUT_ORDER=0: the descent
kernels, the path the earlier rounds pinned)."""
import numpy as np
import pytest

from helpers import layout_hibf, random_hibf, MERGED

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from tetrex_amd import capi as c
    c.init(0)
    return c


def _spell(v, k=4):
    return "".join("ACDEFGHIKLMNPQRSTVWY"[(int(v) >> (5 * (k - 1 - j))) & 31] for j in range(k))


def _queries(values, k=4):
    ok = [v for b in range(0, len(values), max(1, len(values) // 40)) for v in values[b][:2] if all(((int(v) >> (5 * j)) & 31) < 20 for j in range(k))]
    planted = [_spell(v, k) for v in ok]
    qs = ["LMK.{1,3}A[DE]..GK", "WKL..[LIVM]D.[FY]", "LMKA.C.E.GH", "KRK[RK]{2,3}.DE", "CLM.{2,4}C...[LIVMFYWC]", "LMA(E|Q)GLYN", "A.CD", "K[RK]DE"]
    qs += planted[:16] + [p[0] + "." + p[2:] for p in planted[:8]] + [p[:2] + "[" + "".join(sorted(set(p[2] + "AK"))) + "]" + p[3:] for p in planted[8:16]]
    qs += [p[:1] + ".." + p[3:] for p in planted[16:22]] + [p + ".{0,2}" + q for p, q in zip(planted[22:26], planted[26:30])]
    return qs


@pytest.mark.parametrize("tree", ["random-2", "random-3", "random-4", "layout-64", "layout-16-deep", "layout-200-h3"])
def test_queries_on_general_trees_in_layout_order(capi, oracle, monkeypatch, tree):
    monkeypatch.setenv("TETREX_DENSE_EVIDENCE", "dense")
    kind, _, arg = tree.partition("-")
    if kind == "random":
        ox, descs, values = random_hibf(oracle, 40 + int(arg), user_bins=420, levels=int(arg), n_values=60)
        ub = 420
    elif arg == "64":
        ox, descs, values = layout_hibf(oracle, 5, user_bins=3000, tmax=64, n_values=30)
        ub = 3000
    elif arg == "16-deep":
        ox, descs, values = layout_hibf(oracle, 6, user_bins=900, tmax=16, n_values=30, direct=3)  # four levels
        ub = 900
    else:
        ox, descs, values = layout_hibf(oracle, 7, user_bins=5000, tmax=200, h=3, n_values=25)
        ub = 5000
    qs = _queries(values)
    wants = [ox.query(q, with_stats=True) for q in qs]
    ix = capi.Index.upload_hibf(ub, descs)
    assert ix.supports_dense() == 2  # fused steps: the index has a layout order
    results = {}
    # (the index's table of all k-mers' masks would take the dense steps over — these trees are small enough for it — so it is
    # switched off for the ways that name a path, and gets a way of its own, last: once built it stays with the index)
    # ("layout": the rows of plain k-mers by one wave per k-mer — hibf_fused_kernel<G, LAYOUT> —, "layout-levels": level by level)
    # ("layout-lds": the same kernel with the k-mer's row kept in LDS and written out at the end, TXQ_HIBF_LAYOUT_DIRECT=0 — by default its
    # lanes store their words straight into the row in HBM)
    for way in ("layout", "layout-lds", "layout-levels", "layout-blocks", "layout-tracked", "user-order", "table", "table-tracked"):
        monkeypatch.setenv("TXQ_KMER_TABLE_MB", "512" if way.startswith("table") else "0")
        monkeypatch.setenv("TXQ_KMER_TABLE_MIN", "1")
        monkeypatch.delenv("TETREX_DENSE_MIN", raising=False)
        monkeypatch.delenv("TETREX_DENSE_SPARSE_BELOW", raising=False)
        monkeypatch.delenv("TETREX_DENSE_TRACKED", raising=False)
        monkeypatch.delenv("TXQ_HIBF_LAYOUT_ORDER", raising=False)
        monkeypatch.delenv("TXQ_HIBF_LAYOUT_FUSED", raising=False)
        monkeypatch.delenv("TXQ_HIBF_LAYOUT_DIRECT", raising=False)
        if way == "layout-lds":
            monkeypatch.setenv("TXQ_HIBF_LAYOUT_DIRECT", "0")
        if way == "layout-levels":
            monkeypatch.setenv("TXQ_HIBF_LAYOUT_FUSED", "0")
        if way not in ("layout", "layout-lds", "layout-levels"):
            monkeypatch.setenv("TETREX_DENSE_MIN", "2")
            monkeypatch.setenv("TETREX_DENSE_SPARSE_BELOW", "2")
        if way in ("layout-tracked", "table-tracked"):
            monkeypatch.setenv("TETREX_DENSE_TRACKED", "1")
        if way == "user-order":
            monkeypatch.setenv("TXQ_HIBF_LAYOUT_ORDER", "0")
        got, status, stats = ix.query_masks(qs, False, 4)
        results[way] = got
        hits = 0
        for q, g, st, (want, ost) in zip(qs, got, status, wants):
            assert st == 0, q
            if not ost["quirk_merges"]:
                assert np.array_equal(g, want), (q, way)
                hits += int(want.any())
        assert hits >= 10, way
        if way not in ("layout", "layout-lds", "layout-levels"):
            assert stats["dense_ops"] > 0
        if way in ("layout-tracked", "table-tracked"):
            assert stats["tracked_queries"] > 0
    for way, got in results.items():
        assert np.array_equal(got, results["user-order"]), way
    # plain probes keep user-bin order (the public contract of txq_probe)
    kmers = np.concatenate([v[:1] for v in values[:500]] + [np.random.default_rng(1).integers(0, 1 << 20, size=500, dtype=np.uint64)])
    assert np.array_equal(ix.probe(kmers), ox.probe(kmers))
    ix.free()


def test_a_65536_bin_three_level_tree(capi, oracle, monkeypatch):
    """BASELINE configs[4]'s size on a tree as a layout algorithm would shape it: 65 536 user bins, at most 64 technical bins
    per IBF -> three levels, some 3 700 IBFs, user bins scattered over the leaves.  Queries in layout order against the
    oracle, and against the user-order run."""
    monkeypatch.setenv("TETREX_DENSE_EVIDENCE", "dense")
    ox, descs, values = layout_hibf(oracle, 9, user_bins=65536, tmax=64, n_values=12)
    qs = _queries(values)[:40]
    wants = [ox.query(q, with_stats=True) for q in qs]
    ix = capi.Index.upload_hibf(65536, descs)
    assert ix.supports_dense() == 2
    got, status, stats = ix.query_masks(qs, False, 4)
    monkeypatch.setenv("TXQ_HIBF_LAYOUT_ORDER", "0")
    ref, status0, _ = ix.query_masks(qs, False, 4)
    assert list(status) == list(status0) and np.array_equal(got, ref)
    hits = 0
    for q, g, st, (want, ost) in zip(qs, got, status, wants):
        assert st == 0, q
        if not ost["quirk_merges"]:
            assert np.array_equal(g, want), q
            hits += int(want.any())
    assert hits >= 10
    ix.free()


def _value(s):
    v = 0
    for c in s:
        v = (v << 5) | "ACDEFGHIKLMNPQRSTVWY".index(c)
    return v


@pytest.mark.parametrize("way", ["default", "enumerated", "level-kernels", "tracked", "untracked", "two-shards"])
def test_a_motif_whose_kmers_lie_in_different_parts_of_a_split_user_bin(capi, oracle, monkeypatch, way):
    """A user bin that the layout splits over several technical bins holds a k-mer when ANY part does, and the collector
    combines masks per USER bin (reference include/index_hibf.h:132-147).  In layout order a split bin is therefore one bit,
    its first part's: here the k-mers of `LMKACD` are planted in the FIRST, the LAST and a MIDDLE value of every seventh user
    bin — different parts wherever such a bin is split —, so a row ANDed part by part loses those bins (found by
    tools/gpu_regex_fuzz.py, seed 52: `((...){2,3})+` on this tree)."""
    def plant(values):
        for ub in range(0, len(values), 7):
            values[ub][0], values[ub][-1], values[ub][len(values[ub]) // 2] = _value("LMKA"), _value("KACD"), _value("MKAC")
    ox, descs, values = layout_hibf(oracle, 5, user_bins=900, tmax=32, n_values=30, plant=plant)
    split = set()
    for d in descs:
        ubs = [int(u) for u in d["tb_to_user"] if int(u) != MERGED]
        split |= {u for u in ubs if ubs.count(u) > 1}
    assert len([u for u in range(0, 900, 7) if u in split]) >= 10
    env = {"enumerated": {"TETREX_DENSE": "0"}, "level-kernels": {"TXQ_HIBF_LAYOUT_FUSED": "0"}, "tracked": {"TETREX_DENSE_TRACKED": "1"},
           "untracked": {"TETREX_DENSE_TRACKED": "-1"}}.get(way, {})
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    qs = ["LMKAC", "LMKACD", "MKACD", "L.KAC", "LMKA[CD]D", "LM..CD", "L.{1,2}KACD", "((...){2,3})+", "(LMKA|A)C.?D"]
    if way == "two-shards":
        shards = [capi.Index.upload_hibf(900, descs, shard_rank=r, n_shards=2, subtrees=True) for r in range(2)]
        got, status, _ = capi.query_masks_sharded(shards, qs, False, 4)
        for ix in shards:
            ix.free()
    else:
        ix = capi.Index.upload_hibf(900, descs)
        got, status, _ = ix.query_masks(qs, False, 4)
        ix.free()
    planted = 0
    for q, g, st in zip(qs, got, status):
        want = ox.expected_mask(q)[0]
        assert st == 0 and np.array_equal(g, want), (way, q, np.nonzero(np.unpackbits((g ^ want).view(np.uint8), bitorder="little"))[0][:8])
        planted += int(sum((int(want[u >> 6]) >> (u & 63)) & 1 for u in range(0, 900, 7) if u in split))
    assert planted >= 30  # the answers do hold split bins that only the OR of their parts puts there


def test_a_five_level_tree_is_not_taken_in_layout_order(capi, oracle, monkeypatch):
    """The fused layout-order steps follow at most three ancestors (txq_internal.hpp kMaxVDepth): a deeper tree must not get a
    layout order at upload — its sessions run in user-bin order through the descent, and still equal the oracle."""
    monkeypatch.setenv("TETREX_DENSE_EVIDENCE", "dense")
    monkeypatch.setenv("TXQ_KMER_TABLE_MB", "0")
    ox, descs, values = random_hibf(oracle, 77, user_bins=300, levels=5, n_values=40)
    depth = {0: 1}
    for i, d in enumerate(descs):  # (parents come before their children in random_hibf's numbering)
        for nxt, ub in zip(d["next_ibf_id"], d["tb_to_user"]):
            if int(ub) == MERGED:
                depth[int(nxt)] = depth[i] + 1
    assert max(depth.values()) == 5
    ix = capi.Index.upload_hibf(300, descs)
    assert ix.supports_dense() != 2
    qs = _queries(values)
    got, status, stats = ix.query_masks(qs, False, 4)
    for q, g, st in zip(qs, got, status):
        assert st == 0, q
        want, ost = ox.query(q, with_stats=True)
        if not ost["quirk_merges"]:
            assert np.array_equal(g, want), q
    ix.free()


@pytest.mark.parametrize("tree,R", [("random-3", 2), ("random-4", 3), ("layout-64", 8), ("layout-16-deep", 3), ("layout-64", 5)])
def test_general_trees_shard_by_sub_trees(capi, oracle, monkeypatch, tree, R):
    """txq_index_upload_subtrees (BASELINE configs[4] on several GPUs for a tree as seqan::hibf lays it out, reference
    include/index_hibf.h:114-147): the root is replicated, its sub-trees are dealt over the shards, every shard works in layout
    order on its own part of the tree and emits full-width masks that are ORed (info.join_or; split bins may straddle shards).
    Here all shards share the one GPU of the box.  The joined masks of whole queries (one expansion driving all shards:
    txe_query_masks_sharded) and the ORed plain probes must equal the oracle's and the unsharded index's."""
    monkeypatch.setenv("TETREX_DENSE_EVIDENCE", "dense")
    monkeypatch.setenv("TXQ_KMER_TABLE_MB", "0")
    kind, _, arg = tree.partition("-")
    if kind == "random":
        ox, descs, values = random_hibf(oracle, 50 + int(arg), user_bins=420, levels=int(arg), n_values=60)
        ub = 420
    elif arg == "64":
        ox, descs, values = layout_hibf(oracle, 5, user_bins=3000, tmax=64, n_values=30)
        ub = 3000
    else:
        ox, descs, values = layout_hibf(oracle, 6, user_bins=900, tmax=16, n_values=30, direct=3)
        ub = 900
    qs = _queries(values)
    one = capi.Index.upload_hibf(ub, descs)
    ref, status_ref, _ = one.query_masks(qs, False, 4)
    shards = [capi.Index.upload_hibf(ub, descs, shard_rank=r, n_shards=R, subtrees=True) for r in range(R)]
    W = (ub + 63) // 64
    for s_ in shards:
        assert s_.info.join_or == 1 and int(s_.info.shard_word0) == 0 and s_.shard_words == W and s_.info.n_shards == R
    assert sum(int(s_.info.n_ibf) - 1 for s_ in shards) == len(descs) - 1  # every sub-tree IBF lives in exactly one shard
    for way in ("layout", "user-order"):
        if way == "user-order":
            monkeypatch.setenv("TXQ_HIBF_LAYOUT_ORDER", "0")
        full, status, stats = capi.query_masks_sharded(shards, qs, False, 4)
        assert list(status) == list(status_ref) and np.array_equal(full, ref), way
    hits = 0
    for q, g, st in zip(qs, ref, status_ref):
        assert st == 0, q
        want, ost = ox.query(q, with_stats=True)
        if not ost["quirk_merges"]:
            assert np.array_equal(g, want), q
            hits += int(want.any())
    assert hits >= 10
    # plain probes: every shard reports the user bins of its own sub-trees, the OR is membership_for of the whole tree
    kmers = np.concatenate([v[:1] for v in values[:400]] + [np.random.default_rng(2).integers(0, 1 << 20, size=300, dtype=np.uint64)])
    got = np.zeros((kmers.size, W), dtype=np.uint64)
    for s_ in shards:
        got |= s_.probe(kmers)
    assert np.array_equal(got, ox.probe(kmers))
    for s_ in shards + [one]:
        s_.free()


def test_a_regular_tree_keeps_its_column_shards(capi, oracle):
    """txq_index_upload_subtrees on the layout `tetrex index` writes (regular two-level tree): mask columns, as txq_index_upload."""
    from helpers import regular_hibf
    rng = np.random.default_rng(4)
    ox, descs, values = regular_hibf(oracle, 1024, 16, 40, lambda b: rng.integers(0, 1 << 20, size=40, dtype=np.uint64), h=2)
    shards = [capi.Index.upload_hibf(1024, descs, shard_rank=r, n_shards=2, subtrees=True) for r in range(2)]
    assert [int(s_.info.join_or) for s_ in shards] == [0, 0] and [s_.shard_words for s_ in shards] == [8, 8]
    kmers = np.concatenate([v[:1] for v in values[:200]])
    want = ox.probe(kmers)
    assert np.array_equal(np.concatenate([s_.probe(kmers) for s_ in shards], axis=1), want)
    for s_ in shards:
        s_.free()
