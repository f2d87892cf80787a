"""End-to-end candidate-bin masks on the GPU: regex -> C++ host compiler -> mask-DAG blob ->
txq_run_programs (probe + executor kernels) against the CPU oracle's restatement of
query.cpp -> collect() -> bulk_contains on the same index.  Bit-exact."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
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


def _oracle_index(oracle, bins, m, h, k, dna, per_bin, seed, reduction=0):
    ox = oracle.Index.ibf(bins, m, h, dna=dna, k=k, reduction=reduction)
    rng = np.random.default_rng(seed)
    bits = (2 if dna else 5) * k
    for b in range(bins):
        ox.emplace(rng.integers(0, 1 << min(bits, 62), size=per_bin, dtype=np.uint64), b)
    return ox


def _run(capi, host, oracle, ox, queries, dna, k, reduction=0, shards=(1,)):
    sh = ox.shape()
    blob, status, _ = host.compile_batch(queries, dna, k, reduction, ox.bins)
    wants = []
    for q, st in zip(queries, status):
        try:
            wants.append(ox.expected_mask(q)[0])
        except Exception:
            assert st != 0
            wants.append(None)
    checked = 0
    for R in shards:
        for r in range(R):
            ix = capi.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words(), shard_rank=r, n_shards=R)
            lo, nw = int(ix.info.shard_word0), ix.shard_words
            got = ix.run_programs(blob, len(queries))
            for q, g, w in zip(queries, got, wants):
                if w is not None:
                    assert np.array_equal(g, w[lo:lo + nw]), (q, R, r)
                    checked += 1
            ix.free()
    return checked


def test_peptide_queries_on_1024_bins(capi, host, oracle):
    ox = _oracle_index(oracle, bins=1024, m=4099, h=3, k=4, dna=False, per_bin=1500, seed=1)
    qs = PEPTIDE_QUERIES + random_prosite_motifs(60, 7, wildcard=0.05, ranges=0.02)
    assert _run(capi, host, oracle, ox, qs, False, 4, shards=(1, 4)) > 300


def test_dna_queries_various_widths(capi, host, oracle):
    for bins, m, k, per_bin in ((5, 106, 3, 12), (70, 257, 3, 8), (300, 4099, 5, 300)):
        ox = _oracle_index(oracle, bins=bins, m=m, h=3, k=k, dna=True, per_bin=per_bin, seed=bins)
        assert _run(capi, host, oracle, ox, DNA_QUERIES, True, k) > 10


def test_reduced_alphabet_queries(capi, host, oracle):
    for red in (1, 2):
        ox = _oracle_index(oracle, bins=256, m=8191, h=2, k=5, dna=False, per_bin=1500, seed=red, reduction=red)
        qs = ["LMA(E|Q)GLYN", "LMAEGLYNK", "W[LIVM]DVFYLK", "LMAE(GL|YN)KRDE", "KRDEGLYNLMA"]
        _run(capi, host, oracle, ox, qs, False, 5, red)


def test_config1_readme_example_on_gpu(capi, host, oracle, golden):
    """BASELINE configs[0]: data/dna_example_split, k=3, A(C+|G+)T -> bins {0,1,3} (README.md:43-51)."""
    files = [os.path.join(GOLDEN, "dna_example_split", "sequence%d.fa" % i) for i in range(1, 6)]
    per_bin = [sum((host.record_values(s, 3, dna=True, wraparound=True) for _, s in oracle.read_fasta(f)), []) for f in files]
    m = oracle.compute_bitcount(max(map(len, per_bin)), 0.05)
    assert m == 106
    ix = capi.Index.create_ibf(5, m, 3)
    for b, v in enumerate(per_bin):
        dv = capi.DeviceBuffer.from_numpy(np.array(v, dtype=np.uint64))
        db = capi.DeviceBuffer.from_numpy(np.full(len(v), b, dtype=np.uint32))
        ix.emplace_device(dv.ptr, db.ptr, len(v))
    capi.synchronize()
    blob, status, _ = host.compile_batch(["A(C+|G+)T"], True, 3, 0, 5)
    got = ix.run_programs(blob, 1)[0]
    assert [b for b in range(5) if (int(got[0]) >> b) & 1] == golden("config1_masks.json")["quirk"]["candidate_bins"]
    ix.free()


def test_query_on_reference_built_fixture(capi, host, oracle, golden):
    """AC+G on the bits of test/data/ibf_idx.ibf -> candidate mask 0b11 (test/cli/kbioreg_test.cpp:66-79
    expects both records of file1 after verification; bin 1 is a Bloom false positive)."""
    fx = oracle.read_legacy_fixture(os.path.join(GOLDEN, "ibf_idx.ibf"))
    ix = capi.Index.upload_ibf(2, 64, 3, fx["words"])
    blob, _, _ = host.compile_batch(["AC+G"], True, 3, 0, 2)
    assert int(ix.run_programs(blob, 1)[0][0]) == 0b11
    ix.free()


def test_hibf_queries(capi, host, oracle):
    from helpers import random_hibf
    ox, descs, values = random_hibf(oracle, 21, user_bins=300, levels=3, n_values=60)
    # motifs spelled from inserted k-mers of some bins, so several masks are non-trivial
    def spellable(v):
        return all(((int(v) >> s) & 31) < 20 for s in (15, 10, 5, 0))

    def spell(v):
        return "".join("ACDEFGHIKLMNPQRSTVWY"[(int(v) >> s) & 31] for s in (15, 10, 5, 0))
    qs = [spell(next(v for v in values[b] if spellable(v))) for b in range(0, 300, 17)] + ["LMA(E|Q)GLYN", "A.CD", "K[RK]DE"]
    blob, status, _ = host.compile_batch(qs, False, 4, 0, 300)
    ix = capi.Index.upload_hibf(300, descs)
    got = ix.run_programs(blob, len(qs))
    hits = 0
    for q, g in zip(qs, got):
        want = ox.query(q)
        assert np.array_equal(g, want), q
        hits += int(want.any())
    assert hits >= 5
    ix.free()


def test_motifs_as_one_text_equal_motifs_as_an_array(capi, oracle):
    """txe_query_masks_text (include/txh.h: one motif per line, as `tetrex query` reads a batch) against txe_query_masks on the
    same motifs — Index.query_masks takes the text route whenever no motif holds a line break —, with empty lines and a
    motif that fails to parse in the batch; a text of the wrong number of lines is refused."""
    import ctypes as C
    ox = _oracle_index(oracle, bins=200, m=4099, h=3, k=4, dna=False, per_bin=900, seed=5)
    sh = ox.shape()
    ix = capi.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words())
    qs = ["LMAEGLYN", "", "A.C[DE]F", "A{2,}", "W.{1,2}K", ""]
    got, status, _ = ix.query_masks(qs, False, 4)                       # the text
    got_arr, status_arr, _ = ix.query_masks(qs + ["A\nC"], False, 4)    # (a line break in a motif: the array of C strings)
    assert status == status_arr[:len(qs)] and np.array_equal(got, got_arr[:len(qs)])
    assert status[3] != 0 and status[0] == 0 and isinstance(status, list)
    for q, g, st in zip(qs, got, status):
        if st == 0 and q:
            assert np.array_equal(g, ox.query(q)), q
    Lq = capi._query_lib()
    masks = np.zeros((3, ix.shard_words), dtype=np.uint64)
    st3 = (C.c_int * 3)()
    for text in (b"ACDE\nLMAE", b"ACDE\nLMAE\nWKWK\nAAAA"):  # two lines, four lines: not the three stated
        rc = Lq.txe_query_masks_text(ix._h, 0, 4, 0, text, len(text), 3, 0, masks.ctypes.data_as(C.POINTER(C.c_uint64)), st3, None)
        assert rc == -1 and b"number of motifs" in Lq.txe_last_error()
    rc = Lq.txe_query_masks_text(ix._h, 0, 4, 0, b"ACDE\nLMAE\nWKWK\n", 15, 3, 0, masks.ctypes.data_as(C.POINTER(C.c_uint64)), st3, None)
    assert rc == 0 and np.array_equal(masks[1], ox.query("LMAE"))      # (a final line break is optional)
    ix.free()
