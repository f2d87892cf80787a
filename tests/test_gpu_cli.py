"""The `tetrex` command line on the GPU box, in the style of the reference's CLI tests
(test/cli/cli_test.hpp: spawn the binary, capture stdout/stderr).  Expectations come from the
reference's README (README.md:43-51) and its test/cli/kbioreg_test.cpp:66-79."""
import glob
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
TETREX = os.path.join(ROOT, "bin", "tetrex")


def run(*args, cwd=None, stdin=None):
    r = subprocess.run([TETREX, *args], capture_output=True, text=True, cwd=cwd, input=stdin, timeout=300)
    return r.returncode, r.stdout, r.stderr


def rows(stdout):
    return [tuple(line.split("\t")) for line in stdout.splitlines() if line]


@pytest.fixture(scope="module")
def toy(tmp_path_factory):
    d = tmp_path_factory.mktemp("toy")
    files = sorted(glob.glob(os.path.join(GOLDEN, "dna_example_split", "*.fa")))
    out = {}
    for flavour, flag in (("ibf", ["-i"]), ("hibf", [])):
        rc, so, se = run("index", "-n", "-k", "3", *flag, str(d / flavour), *files)
        assert rc == 0, se
        assert "Indexed 5 sequences across 5 bins." in se and "DONE" in se
        out[flavour] = str(d / (flavour + ".ibf"))
    return out


@pytest.mark.parametrize("flavour", ["ibf", "hibf"])
def test_readme_example(toy, flavour):
    """tetrex query test.ibf "A(C+|G+)T": Sequence1 ACT x3, Sequence2 ACT + AGT, Sequence4 ACCCT."""
    rc, so, se = run("query", "-v", toy[flavour], "A(C+|G+)T")
    assert rc == 0, se
    fwd = [(os.path.basename(r[0]), r[1], r[2]) for r in rows(so) if "REVERSE" not in r[3]]
    assert fwd == [("sequence1.fa", ">Sequence1", "ACT")] * 3 + [("sequence2.fa", ">Sequence2", "ACT"),
                                                                 ("sequence2.fa", ">Sequence2", "AGT"),
                                                                 ("sequence4.fa", ">Sequence4", "ACCCT")]
    # start,end columns of the current format (src/query.cpp:212-216)
    assert [r[3] for r in rows(so) if "sequence1" in r[0] and "REVERSE" not in r[3]] == ["0,3", "4,7", "8,11"]
    assert "Narrowed Search to 3 possible bins" in se and "Query Time: " in se


def test_inspect(toy):
    rc, so, se = run("inspect", toy["ibf"])
    assert "INDEX TYPE: IBF" in so and "BIN COUNT (BFs): 5" in so and "BIN SIZE (bits): 106" in so
    assert "HASH COUNT (hash functions): 3" in so and "KMER LENGTH (bases): 3" in so
    rc, so, se = run("inspect", toy["hibf"])
    assert "INDEX TYPE: HIBF" in so and "FALSE POSITIVE RATE: 0.05" in so and so.count("\t- ") == 5


def test_index_reproduces_the_reference_built_fixture(tmp_path, oracle):
    """`tetrex index` (bits set by the GPU emplace kernel) on file1.fa/file2.fa with the fixture's
    shape gives the byte-identical bit matrix of the reference-built test/data/ibf_idx.ibf, and the
    query of test/cli/kbioreg_test.cpp:66-79 finds Snippet1.1 ACCG and Snippet1.2 ACG only."""
    from tetrex_amd import host
    fx = oracle.read_legacy_fixture(os.path.join(GOLDEN, "ibf_idx.ibf"))
    fpr = "0.022"  # 8 k-mers in the larger bin (file2) -> ceil(-8 ln p / ln^2 2) = 64 rows, the fixture's bin size
    assert oracle.compute_bitcount(8, float(fpr)) == 64
    rc, so, se = run("index", "-n", "-i", "-k", "3", "-p", fpr, "--no-wraparound", str(tmp_path / "fx"),
                     os.path.join(GOLDEN, "file1.fa"), os.path.join(GOLDEN, "file2.fa"))
    assert rc == 0, se
    mine = host.IndexFile.load(str(tmp_path / "fx.ibf"))
    assert mine.describe()["ibfs"][0] == dict(bins=2, tech_bins=64, bin_size=64, hash_shift=57, bin_words=1, hash_funs=3)
    assert np.array_equal(mine.words(), fx["words"])
    rc, so, se = run("query", str(tmp_path / "fx.ibf"), "AC+G")
    got = [(r[1], r[2]) for r in rows(so) if "REVERSE" not in r[3]]
    assert got == [(">Snippet1.1", "ACCG"), (">Snippet1.2", "ACG")]


def test_query_reads_regex_from_stdin_and_writes_to_file(toy, tmp_path):
    dest = tmp_path / "hits.tsv"
    rc, so, se = run("query", "-o", str(dest), toy["ibf"], "-", stdin="A(C+|G+)T\n")
    assert rc == 0
    assert len([l for l in dest.read_text().splitlines() if l]) == 6
    assert all("REVERSE STRAND HIT" in l for l in so.splitlines() if l)  # reverse hits always go to stdout


def test_motif_file_mode(toy, tmp_path):
    motifs = tmp_path / "motifs.tsv"
    motifs.write_text("m1\tA(C+|G+)T\nm2\tCCCGTACCC\n\nm3\tTTTTTT\n")
    rc, so, se = run("query", "-f", toy["ibf"], str(motifs), cwd=str(tmp_path))
    assert rc == 0, se
    assert (tmp_path / "m1.tsv").exists() and (tmp_path / "m2.tsv").exists()
    assert len((tmp_path / "m1.tsv").read_text().splitlines()) == 6
    m2 = [l.split("\t")[1] for l in (tmp_path / "m2.tsv").read_text().splitlines()]
    assert m2 == [">Sequence4", ">Sequence5"]
    lines = [l for l in se.splitlines() if l.startswith("m")]
    assert lines[0].startswith("m1\tBin Count: 3\tQuery Time: ") and lines[2].startswith("m3\tBin Count: ")


def test_peptide_index_and_query(tmp_path):
    rng = np.random.default_rng(0)
    aa = list("ACDEFGHIKLMNPQRSTVWY")
    files = []
    for b in range(70):
        seqs = ["".join(rng.choice(aa, size=200)) for _ in range(5)]
        if b in (7, 33):
            seqs[2] = seqs[2][:50] + "LMAEGLYN" + seqs[2][58:]
        if b == 50:
            seqs[1] = seqs[1][:10] + "LMAQGLYN" + seqs[1][18:]
        p = tmp_path / ("bin%02d.fa" % b)
        p.write_text("".join(">s%d_%d some comment\n%s\n" % (b, i, s) for i, s in enumerate(seqs)))
        files.append(str(p))
    for flag in (["-i"], []):
        rc, so, se = run("index", "-k", "4", *flag, str(tmp_path / "pep"), *files)
        assert rc == 0 and "Indexed 350 sequences across 70 bins." in se
        rc, so, se = run("query", "-v", "-t", "4", str(tmp_path / "pep.ibf"), "LMA(E|Q)GLYN")
        got = sorted((os.path.basename(r[0]), r[2]) for r in rows(so))
        assert got == [("bin07.fa", "LMAEGLYN"), ("bin33.fa", "LMAEGLYN"), ("bin50.fa", "LMAQGLYN")]
    rc, so, se = run("query", "-c", str(tmp_path / "pep.ibf"), "LMAEG:GLYN")
    assert sorted(os.path.basename(r[0]) for r in rows(so)) == ["bin07.fa", "bin33.fa"]


def test_bad_index_path_and_bad_query(toy):
    rc, so, se = run("query", "/nonexistent.ibf", "ACGT")
    assert "Filepath to (H)IBF Index not valid" in se
    rc, so, se = run("query", toy["ibf"], "A{2,}")
    assert rc != 0 and "not searchable" in se


def test_stats_flag_prints_one_json_line(toy):
    """-S/--stats (an extension of this build): the candidate-mask stage in numbers, on stderr."""
    import json
    rc, so, se = run("query", "-S", toy["ibf"], "A(C+|G+)T")
    assert rc == 0, se
    lines = [ln for ln in se.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, se
    st = json.loads(lines[0])
    assert st["queries"] == 1 and st["bins"] == 5 and st["stages"] >= 1 and st["ops"] > 0 and st["kmer_probes"] > 0
    assert st["mask_seconds"] >= st["execute_seconds"] >= 0


def test_megabyte_records_with_quantifier_motifs(tmp_path):
    """with a backtracking matcher a chromosome-length record and a `+` / {m,n} motif overflowed the stack or took
    exponential time in the verification stage.  Two bins of one 1.5 MB DNA record each; motifs with + * {m,n}; both strands."""
    rng = np.random.default_rng(21)
    files = []
    for b in range(2):
        seq = "".join(rng.choice(list("ACGT"), size=1_500_000))
        if b == 1:
            seq = seq[:700000] + "GATTACCCCCCCCCCCCCCCCCCCCA" + seq[700026:]
        p = tmp_path / ("chr%d.fa" % b)
        p.write_text(">chr%d\n%s\n" % (b, seq))
        files.append(str(p))
    rc, so, se = run("index", "-n", "-i", "-k", "8", str(tmp_path / "big"), *files)
    assert rc == 0, se
    rc, so, se = run("query", "-v", str(tmp_path / "big.ibf"), "GATTAC{10,30}A")
    assert rc == 0, se
    fwd = [r for r in rows(so) if "REVERSE" not in r[3]]
    assert [(os.path.basename(r[0]), r[2], r[3]) for r in fwd] == [("chr1.fa", "GATTACCCCCCCCCCCCCCCCCCCCA", "700000,700026")]
    rc, so, se = run("query", str(tmp_path / "big.ibf"), "ACGT(AC)+GTTT(G|T)+AAC")
    assert rc == 0, se
    for r in rows(so):
        assert r[2].startswith("ACGTAC") and r[2].endswith("AAC")


def test_hibf_written_by_tetrex_index_takes_the_fused_tree_steps(tmp_path):
    """`tetrex index` deals the user bins over word-aligned children of equal rows (host/device_index.cpp), so the device
    recognises the tree as regular with uniform children and runs wildcard motifs as fused dense steps on the
    interleaved children.  The verified matches equal those found through a flat IBF of the same bins."""
    import json
    from tetrex_amd import host
    rng = np.random.default_rng(3)
    aa = list("ACDEFGHIKLMNPQRSTVWY")
    files = []
    for b in range(200):
        seqs = ["".join(rng.choice(aa, size=150)) for _ in range(3)]
        if b % 37 == 5:
            seqs[1] = seqs[1][:40] + "LMKWACDEQGHK" + seqs[1][52:]
        p = tmp_path / ("bin%03d.fa" % b)
        p.write_text("".join(">s%d_%d\n%s\n" % (b, i, s) for i, s in enumerate(seqs)))
        files.append(str(p))
    lst = tmp_path / "bins.lst"
    lst.write_text("\n".join(files) + "\n")
    out = {}
    for name, flag in (("flat", ["-i"]), ("tree", [])):
        rc, so, se = run("index", "-k", "4", *flag, str(tmp_path / name), str(lst))
        assert rc == 0 and "across 200 bins." in se, se
    img = host.IndexFile.load(str(tmp_path / "tree.ibf")).describe()
    assert img["is_hibf"] and [f["bins"] for f in img["ibfs"]] == [4, 64, 64, 64, 8]
    assert len({f["bin_size"] for f in img["ibfs"][1:]}) == 1
    env = dict(os.environ, TXQ_TRACE="1", TETREX_DENSE_EVIDENCE="dense")  # (600 residues per bin: left alone, the expansion learns that states thin out)
    fused = 0
    for q in ("LMK.AC.{0,2}EQ[GA]HK", "LMK..CDEQGHK", "LMKW.{2,3}EQGHK", "W.C[DE]{1,2}.GHK"):
        for name in ("flat", "tree"):
            r = subprocess.run([TETREX, "query", "-S", "-t", "4", str(tmp_path / (name + ".ibf")), q], capture_output=True, text=True, env=env, timeout=300)
            assert r.returncode == 0, r.stderr
            out[name] = sorted((os.path.basename(x[0]), x[1], x[2], x[3]) for x in rows(r.stdout))
            st = json.loads([ln for ln in r.stderr.splitlines() if ln.startswith("{")][0])
            if name == "tree" and st["dense_ops"]:
                assert "interleaved children" in r.stderr, r.stderr
                fused += 1
        assert out["flat"] == out["tree"] and len(out["flat"]) >= 5, q
    assert fused >= 2


@pytest.mark.parametrize("dna", [False, True], ids=["peptides", "dna"])
def test_motif_batches_are_verified_bin_major_with_the_same_files(tmp_path, dna):
    """`tetrex query -f`: the candidate bins of ALL motifs are read once each and every motif that selected a bin runs over its
    records (host/verify.cpp verify_batch) — the reference verifies motif by motif (include/query.h:329-346 over :126-138).
    The result files, the reverse-strand rows on stdout and the per-motif log lines must be what the motif-by-motif run
    (TETREX_VERIFY_PER_MOTIF=1) writes, byte for byte, with one thread and with several."""
    rng = np.random.default_rng(5)
    letters = list("ACGT") if dna else list("ACDEFGHIKLMNPQRSTVWY")
    planted = ["ACGTTGCAAC", "GGATCCAT", "TTGACAGCTAGC"] if dna else ["LMAEGLYN", "WKLPDSFY", "CAAHKCLLMH"]
    files = []
    for b in range(48):
        seqs = ["".join(rng.choice(letters, size=300)) for _ in range(6)]
        for j, w in enumerate(planted):
            if (b + j) % 7 == 0:
                seqs[j] = seqs[j][:40 + b] + w + seqs[j][40 + b + len(w):]
        # records in 60-column lines like UniProt's FASTA, every third bin gzip-compressed, one without a final newline
        body = "".join(">r%d_%d some description\n%s\n" % (b, i, "\n".join(s_[c:c + 60] for c in range(0, len(s_), 60))) for i, s_ in enumerate(seqs))
        if b == 5:
            body = body[:-1]
        if b % 3 == 0:
            import gzip
            p = tmp_path / ("bin%02d.fa.gz" % b)
            with gzip.open(p, "wt") as f:
                f.write(body)
        else:
            p = tmp_path / ("bin%02d.fa" % b)
            p.write_text(body)
        files.append(str(p))
    k = "6" if dna else "4"
    rc, so, se = run("index", *(["-n"] if dna else []), "-k", k, "-i", str(tmp_path / "ix"), *files)
    assert rc == 0, se
    if dna:
        motifs = ["ACGTTGCAAC", "GGAT.CAT", "TTGACA[GC]CTAGC", "ACG(T|A)TGCA", "GTTGCAACGT", "AAAAAAAAAAAA", "AC+GT", "GGATC{1,2}AT"]
    else:
        motifs = ["LMA(E|Q)GLYN", "WKLPD.FY", "CAAHKC[LI]LMH", "LMAEG", "W.LPDSF", "KCLLMH", "QQQQQQQQ", "LM.EGLY", "AAH.{0,2}CLLM"]
    (tmp_path / "motifs.tsv").write_text("".join("M%d\t%s\n" % (i, m) for i, m in enumerate(motifs)))
    results = {}
    for mode, threads in (("per-motif", "1"), ("bin-major", "1"), ("bin-major", "5")):
        out = tmp_path / (mode + threads)
        out.mkdir()
        env = dict(os.environ)
        if mode == "per-motif":
            env["TETREX_VERIFY_PER_MOTIF"] = "1"
        r = subprocess.run([TETREX, "query", "-f", "-t", threads, str(tmp_path / "ix.ibf"), str(tmp_path / "motifs.tsv")], capture_output=True, text=True,
                           cwd=str(out), env=env, timeout=300)
        assert r.returncode == 0, r.stderr
        files_out = {os.path.basename(p): open(p, "rb").read() for p in sorted(glob.glob(str(out / "*.tsv")))}
        log = [l.split("Query Time")[0] for l in r.stderr.splitlines() if l.startswith("M")]
        results[(mode, threads)] = (files_out, r.stdout, log)
    base = results[("per-motif", "1")]
    assert sum(1 for v in base[0].values() if v) >= 4 and len(base[2]) == len(motifs)
    if dna:
        assert "REVERSE STRAND HIT" in base[1]
    for key, got in results.items():
        assert got[0] == base[0], key
        assert got[1] == base[1], key
        assert got[2] == base[2], key
