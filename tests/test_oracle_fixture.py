"""Pin the CPU oracle against everything the reference's own tree holds for this path.

The oracle restates seqan::hibf arithmetic that is absent from /root/reference, so it is
only trustworthy once it reproduces the reference's binary fixture test/data/ibf_idx.ibf
(built by the reference from file1.fa/file2.fa) bit for bit, plus the golden vectors of
SURVEY.md §8(c).  CPU only.
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN


def test_fixture_sha256_matches_reference_datasources():
    # /root/reference/test/data/datasources.cmake:5-13
    want = {
        "file1.fa": "71e7416fe1d7e10c633f253d30e8bbd27e0b8b7dcd4d982e0958c5ddaf19dc27",
        "file2.fa": "1fef24d4e2ed643a89e6aaab2e355231569bcbc1002c2ed2715d8becf95ebad3",
        "ibf_idx.ibf": "fbcc94558c881c8baf90eb8663288ba738c7e080e771bc60bd5e908bee3b02b4",
    }
    for name, sha in want.items():
        with open(os.path.join(GOLDEN, name), "rb") as f:
            assert hashlib.sha256(f.read()).hexdigest() == sha, name


def test_fixture_header_invariants(oracle):
    fx = oracle.read_legacy_fixture(os.path.join(GOLDEN, "ibf_idx.ibf"))
    assert (fx["bins"], fx["tech_bins"], fx["bin_size"], fx["hash_funs"], fx["k"], fx["molecule"]) == (2, 64, 64, 3, 3, "na")
    assert fx["hash_shift"] == 57 == 64 - int(fx["bin_size"]).bit_length()
    assert fx["bin_words"] == 1 and fx["bits"] == 64 * 64
    assert fx["words"].size == 64


def test_oracle_rebuilds_reference_fixture_bit_for_bit(oracle, golden):
    """hash_and_fit (seeds 0-2), fastrange, row-major interleave, 2-bit code, canonical k-mers."""
    fx = oracle.read_legacy_fixture(os.path.join(GOLDEN, "ibf_idx.ibf"))
    ix = oracle.Index.ibf(2, 64, 3, dna=True, k=3)
    for b, fname in enumerate(["file1.fa", "file2.fa"]):
        for _, seq in oracle.read_fasta(os.path.join(GOLDEN, fname)):
            ix.emplace(oracle.decompose(seq, 3, dna=True, quirk=False), b)
    assert np.array_equal(ix.words(), fx["words"])
    kat = golden("hash_kat.json")["fixture"]
    rows0 = [r for r in range(64) if fx["words"][r] & 1]
    rows1 = [r for r in range(64) if fx["words"][r] & 2]
    assert rows0 == kat["rows_bin0"] and rows1 == kat["rows_bin1"]
    assert not np.any(fx["words"] >> np.uint64(2))


def test_hash_known_answers(oracle, golden):
    kat = golden("hash_kat.json")
    for v, rows in kat["fixture"]["kmer_rows"].items():
        assert oracle.hash_rows(int(v), 64, 3) == rows
    for e in kat["derived_only"]:
        assert oracle.hash_rows(e["value"], e["bin_size"], 5) == e["rows"]
    for e in kat["bitcount"]:
        assert oracle.compute_bitcount(e["n"], e["fpr"]) == e["m"]


def test_bulk_contains_on_reference_bits(oracle, golden):
    """Query AC+G on the reference-built bits: M[ACG]=M[ACC]=0b11, M[CCG]=0b01 -> candidate 0b11."""
    fx = oracle.read_legacy_fixture(os.path.join(GOLDEN, "ibf_idx.ibf"))
    ix = oracle.Index.ibf(2, 64, 3, dna=True, k=3)
    ix.set_words(fx["words"])
    g = golden("config1_masks.json")["fixture_query"]
    enc = {"ACG": 7, "ACC": 5, "CCG": 23}
    for kmer, bits in g["masks"].items():
        assert int(ix.probe([enc[kmer]])[0, 0]) == int(bits, 2), kmer
    mask, stats = ix.query(g["regex"], with_stats=True)
    assert int(mask[0]) == int(g["candidate"], 2)
    assert stats["probes"] == 3 and stats["quirk_merges"] == 0


def test_translate_goldens(oracle, golden):
    for rx, postfix in golden("translate.json")["pairs"]:
        assert oracle.translate(rx) == postfix, rx


def test_translate_errors_give_empty_postfix(oracle):
    # src/utils.cpp:9-13 swallows lexer exceptions and returns ""
    for bad in ["A{2,}", "A[", "A{x}", "A{3,2}", "[]", "A\\"]:
        assert oracle.translate(bad) == ""


def test_encoder_goldens(oracle, golden):
    g = golden("encoders.json")
    fwd, canon = oracle.update_kmers(g["dna_update_kmer_k3"]["symbols"], 3, dna=True)
    assert fwd == g["dna_update_kmer_k3"]["fwd"] and canon == g["dna_update_kmer_k3"]["canon"]
    d = g["dna_decompose_quirk"]
    assert oracle.decompose(d["seq"], d["k"], dna=True, quirk=True) == d["values"]
    assert oracle.decompose(g["aa_k4"]["seq"], 4, dna=False) == g["aa_k4"]["values"]
    fwd, _ = oracle.update_kmers(g["murphy_k5_update"]["seq"], 5, dna=False, reduction=1)
    assert fwd == g["murphy_k5_update"]["fwd"]
    for red, name in enumerate(["base", "murphy", "li"]):
        aamap, _ = oracle.encoder_tables(red)
        for item in g["tables"][name].split():
            assert aamap[ord(item[0])] == int(item[1:]), (name, item)
    rq = g["reduce_query"]
    assert oracle.reduce_alphabet(rq["motif"], 1) == rq["murphy"]
    assert oracle.reduce_alphabet(rq["motif"], 2) == rq["li"]
    for red, name in ((1, "murphy"), (2, "li")):
        for kmer, val in g["reduced_k5_probes"][name].items():
            assert oracle.update_kmers(kmer, 5, dna=False, reduction=red)[0][-1] == val


def test_config2_probe_set(oracle, golden):
    g = golden("config2_kmers.json")
    for kmer, val in g["kmers"].items():
        assert oracle.update_kmers(kmer, 4, dna=False)[0][-1] == val
    # the collector probes exactly those 9 distinct forward k-mers (GLYN re-merges on suffix GLY)
    ix = oracle.Index.ibf(128, 4099, 3, dna=False, k=4)
    rng = np.random.default_rng(1)
    for b in range(128):
        ix.emplace(list(g["kmers"].values()), b)  # keep every path alive so nothing is pruned
    mask, stats = ix.query(g["motif"], with_stats=True)
    assert stats["probes"] == 9
    assert np.all(mask == np.uint64(0xFFFFFFFFFFFFFFFF))


def test_kgraph_config1_matches_hand_trace(oracle, golden):
    g = golden("kgraph_config1.json")
    kg = oracle.kgraph(g["postfix"], g["k"])
    assert kg["labels"] == g["labels"]
    assert sorted(map(tuple, kg["arcs"])) == sorted(map(tuple, g["arcs"]))
    # ranks are a topological order with the start node at 0 and Match last
    for s, t in kg["arcs"]:
        assert kg["ranks"][s] < kg["ranks"][t]
    assert kg["ranks"][0] == 0 and kg["ranks"][-1] == len(kg["labels"]) - 1


def _build_config1(oracle, quirk):
    files = [os.path.join(GOLDEN, "dna_example_split", "sequence%d.fa" % i) for i in range(1, 6)]
    per_bin = []
    for f in files:
        vals = []
        for _, seq in oracle.read_fasta(f):
            vals += oracle.decompose(seq, 3, dna=True, quirk=quirk)
        per_bin.append(vals)
    m = oracle.compute_bitcount(max(len(v) for v in per_bin), 0.05)
    ix = oracle.Index.ibf(5, m, 3, dna=True, k=3)
    for b, vals in enumerate(per_bin):
        ix.emplace(vals, b)
    return ix, m, per_bin


@pytest.mark.parametrize("variant", ["quirk", "plain"])
def test_config1_end_to_end_on_cpu(oracle, golden, variant):
    """BASELINE configs[0]: 5-bin DNA IBF, k=3, query A(C+|G+)T -> candidate bins {0,1,3}."""
    g = golden("config1_masks.json")[variant]
    ix, m, per_bin = _build_config1(oracle, quirk=(variant == "quirk"))
    assert max(len(v) for v in per_bin) == g["n_max"] and m == g["bin_size"]
    for kmer, bits in g["kmer_masks"].items():
        canon = oracle.update_kmers(kmer, 3, dna=True)[1][-1]
        assert int(ix.probe([canon])[0, 0]) == int(bits, 2), kmer
    mask, stats = ix.query("A(C+|G+)T", with_stats=True)
    assert [b for b in range(5) if (int(mask[0]) >> b) & 1] == g["candidate_bins"]
    assert stats["probes"] == 6  # 6 forward k-mers, 3 distinct canonical row sets


def test_regenerable_golden_files_regenerate_identically():
    """tests/golden/regenerate.py --check: hash_kat.json, config1_masks.json and config2_kmers.json come out of an independent
    pure-Python model of SURVEY.md §8(c)'s formulas applied to the reference-held data files, byte for byte as committed."""
    import subprocess
    import sys
    from conftest import GOLDEN
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, "regenerate.py"), "--check"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
