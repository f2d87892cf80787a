"""CPU tests of the C++ host front-end (libtetrex_host.so) against the reference's golden
vectors and against the CPU oracle.  The compiled mask-DAG programs are evaluated here with
numpy over oracle-probed masks, which checks the compiler without a GPU; tests/test_gpu_query.py
runs the same programs on the device."""
import numpy as np
import pytest

from helpers import eval_program, ones_mask
from motifs import PEPTIDE_QUERIES, DNA_QUERIES, random_prosite_motifs


@pytest.fixture(scope="module")
def host():
    from tetrex_amd import host as H
    H.lib()
    return H


def test_translate_goldens(host, golden):
    for rx, postfix in golden("translate.json")["pairs"]:
        assert host.translate(rx) == postfix, rx
    for bad in ["A{2,}", "A[", "A{x}", "A{3,2}", "[]", "A\\", "[^ACDEFGHIKLMNPQRSTVWY]"]:
        assert host.translate(bad) == ""


def test_translate_agrees_with_oracle_on_generated_motifs(host, oracle):
    for rx in random_prosite_motifs(300, 11) + PEPTIDE_QUERIES + DNA_QUERIES + ["A\\|B", "(A", "A)B", "A||B", "a.b*", "A{0}B"]:
        assert host.translate(rx) == oracle.translate(rx), rx


def test_preprocess_reduces_and_trims(host, oracle, golden):
    g = golden("encoders.json")["reduce_query"]
    assert host.preprocess(g["motif"], False, 5, 1)[0] == g["murphy"]
    assert host.preprocess(g["motif"], False, 5, 2)[0] == g["li"]
    for rx in ["^M.[^P]{2}K$", ".*LMAE.+", "[A-Z]LMA.", ".{2,4}LMA[^C]", "LMA$", "^$", "...", "LM.A"]:
        pre, post = host.preprocess(rx, False, 4, 0)
        assert pre == oracle.trim_regex(rx), rx
        assert post == oracle.translate(oracle.trim_regex(rx))
        # DNA queries are translated untouched (include/query.h:90-94)
        assert host.preprocess(rx, True, 4, 0)[0] == rx


def test_encoder_goldens(host, golden):
    g = golden("encoders.json")
    d = g["dna_decompose_quirk"]
    assert host.record_values(d["seq"], d["k"], dna=True, wraparound=True) == d["values"]
    assert host.record_values(g["aa_k4"]["seq"], 4, dna=False) == g["aa_k4"]["values"]
    assert host.record_values("IBKFS", 5, dna=False, reduction=1) == [g["murphy_k5_update"]["fwd"][-1]]


def test_encoder_agrees_with_oracle(host, oracle):
    rng = np.random.default_rng(0)
    for _ in range(200):
        dna = bool(rng.integers(0, 2))
        k = int(rng.integers(2, 33 if dna else 13))
        red = 0 if dna else int(rng.integers(0, 3))
        alpha = "ACGT" if dna else "ABCDEFGHIJKLMNOPQRSTUVWXYZ"
        seq = "".join(rng.choice(list(alpha), size=int(rng.integers(0, 80))))
        wrap = bool(rng.integers(0, 2))
        assert host.record_values(seq, k, dna=dna, reduction=red, wraparound=wrap) == \
            oracle.decompose(seq, k, dna=dna, reduction=red, quirk=wrap), (seq, k, dna, red, wrap)


@pytest.mark.parametrize("k", [2, 3, 4, 6])
def test_kgraph_is_node_for_node_the_oracle_graph(host, oracle, k):
    queries = PEPTIDE_QUERIES + DNA_QUERIES + random_prosite_motifs(60, 5)
    for rx in queries:
        postfix = oracle.translate(rx)
        if not postfix:
            continue
        try:
            want = oracle.kgraph(postfix, k)
        except oracle.OracleError:
            with pytest.raises(host.HostError):
                host.kgraph(postfix, k)
            continue
        got = host.kgraph(postfix, k)
        assert got["labels"] == want["labels"], rx
        assert got["succ"] == [tuple(s) for s in want["succ"]], rx


def _language(labels, succ, members=None, limit=60000):
    """Every residue string a k-graph (a DAG: loops are unrolled) accepts, by walking it from node 0 to the Match node."""
    out, todo = set(), [(0, "")]
    while todo:
        v, text = todo.pop()
        lab = labels[v]
        if lab == 256:  # Match
            out.add(text)
            if len(out) > limit:
                return None  # too many to write out
            continue
        if lab >= 260:
            nexts = [text + c for c in members[v]]
        elif lab < 256 and lab != ord("$"):
            nexts = [text + chr(lab)]
        else:
            nexts = [text]
        for t in {s for s in succ[v] if s >= 0}:
            for nx in nexts:
                todo.append((t, nx))
    return out


@pytest.mark.parametrize("k", [2, 4, 6])
def test_fused_kgraph_accepts_the_same_strings_with_one_node_per_class(host, oracle, k):
    """The graph the expansion works on (unions of single residues fused into class nodes, include/txh.h txh_kgraph_fused)
    against the reference's node-for-node graph: the same language, far fewer nodes."""
    queries = ["A[LIVM]C", "AC(D|E)F", "(A|C|D)(E|F)", "A.C", "[AC]{2}D", "W[DE]{1,2}K", "A([LI]|MC)D", "A[LL]C", "(A|[CD])E",
               "LMA(E|Q)GLYN", "M[KR]+S", "A[DE]*C", "(AC|D)E", "([AC]|[DE])F", "A[CD]?E"]
    queries += [q for q in random_prosite_motifs(40, 9, wildcard=0.1, ranges=0.1) if q.count(".") <= 2 and "{" not in q.replace(".{", "")][:25]
    checked = 0
    for rx in queries:
        postfix = oracle.translate(rx)
        plain = host.kgraph(postfix, k)
        fused = host.kgraph_fused(postfix, k)
        want = _language(plain["labels"], plain["succ"])
        if want is None:
            continue
        checked += 1
        got = _language(fused["labels"], fused["succ"], fused["members"])
        assert got == want, rx
        assert len(fused["labels"]) <= len(plain["labels"]), rx
        for lab, m in zip(fused["labels"], fused["members"]):
            if lab >= 260:
                assert len(m) >= 1 and len(set(m)) == len(m), (rx, m)  # ([LL] is a class of one)
    assert checked >= 25
    g = host.kgraph_fused(oracle.translate("A.[LIVM](E|Q)C"), 4)
    assert [m for m in g["members"] if len(m) > 1] == ["FQLTKPAYRNHGECIVDWSM", "LIVM", "EQ"]
    assert len(g["labels"]) == 7  # start ghost, A, three classes, C, Match


def test_kgraph_config1_hand_trace(host, golden):
    g = golden("kgraph_config1.json")
    kg = host.kgraph(g["postfix"], g["k"])
    assert kg["labels"] == g["labels"]
    arcs = set()
    for u, (a, b) in enumerate(kg["succ"]):
        if a >= 0:
            arcs.add((u, a))
        if b >= 0:
            arcs.add((u, b))
    assert arcs == set(map(tuple, g["arcs"]))


def test_reduced_builder_matches_oracle(host, oracle, golden):
    for postfix in ["II-A-BB|-G-I-F-B-", "JJ-A-BB|-G-J-F-H-", "AB-C-", "AB|C-", "AB-?C-", "AB-C|D-", "AB{2}-C-"]:
        for k in (3, 5):
            try:
                want = oracle.kgraph(postfix, k, reduced=True)
            except oracle.OracleError:
                with pytest.raises(host.HostError):
                    host.kgraph(postfix, k, reduced=True)
                continue
            got = host.kgraph(postfix, k, reduced=True)
            assert got["labels"] == want["labels"] and got["succ"] == [tuple(s) for s in want["succ"]], postfix
    # the redundant union of identical reduced letters collapses to a linear chain
    kg = host.kgraph("II-A-BB|-G-I-F-B-", 5, reduced=True)
    assert [chr(l) for l in kg["labels"] if l < 256] == list("IIABGIFB")


def _index(oracle, bins, m, h, k, dna, per_bin, seed, reduction=0):
    ox = oracle.Index.ibf(bins, m, h, dna=dna, k=k, reduction=reduction)
    rng = np.random.default_rng(seed)
    bits = (2 if dna else 5) * k
    for b in range(bins):
        ox.emplace(rng.integers(0, 1 << min(bits, 62), size=per_bin, dtype=np.uint64), b)
    return ox


def _check_batch(host, ox, queries, dna, k, reduction=0):
    bins = ox.bins
    blob, status, stats = host.compile_batch(queries, dna, k, reduction, bins)
    kmers, progs = host.parse_blob(blob)
    assert len(progs) == len(queries)
    assert len(set(int(x) for x in kmers)) == kmers.size  # deduplicated table
    M = ox.probe(kmers) if kmers.size else np.zeros((0, ox.words_per_mask), dtype=np.uint64)
    ones = ones_mask(bins)
    informative = 0
    for q, (n_slots, ops), st in zip(queries, progs, status):
        try:
            want, ost = ox.query(q, with_stats=True)
        except Exception:
            assert st != 0, q  # the reference has no defined result here; the host must flag it
            continue
        assert st == 0, q
        if ost["quirk_merges"]:
            continue  # reference result depends on hash-map iteration order
        got = eval_program(n_slots, [tuple(int(x) for x in o) for o in ops], M, ones)
        assert np.array_equal(got, want), q
        pop = int(np.unpackbits(want.view(np.uint8)).sum())
        informative += 0 < pop < bins
    return informative


def test_compiled_programs_equal_oracle_collect_peptide(host, oracle):
    ox = _index(oracle, bins=200, m=4099, h=3, k=4, dna=False, per_bin=1500, seed=1)
    qs = PEPTIDE_QUERIES + random_prosite_motifs(40, 2, wildcard=0.05, ranges=0.0)
    assert _check_batch(host, ox, qs, False, 4) >= 10


def test_compiled_programs_equal_oracle_collect_dna(host, oracle):
    ox = _index(oracle, bins=70, m=257, h=3, k=3, dna=True, per_bin=8, seed=3)
    assert _check_batch(host, ox, DNA_QUERIES, True, 3) >= 5
    ox = _index(oracle, bins=130, m=4099, h=2, k=5, dna=True, per_bin=300, seed=4)
    assert _check_batch(host, ox, DNA_QUERIES, True, 5) >= 3


@pytest.mark.parametrize("reduction", [1, 2])
def test_compiled_programs_equal_oracle_collect_reduced(host, oracle, reduction):
    ox = _index(oracle, bins=100, m=8191, h=3, k=5, dna=False, per_bin=1500, seed=5, reduction=reduction)
    qs = ["LMA(E|Q)GLYN", "LMAEGLYNK", "W[LIVM]DVFYLK", "LMAE(GL|YN)KRDE", "KRDEGLYNLMA", "L(MA|KR)EGLYN"]
    _check_batch(host, ox, qs, False, 5, reduction)


def test_one_bin_index_skips_filtering(host, oracle):
    blob, status, _ = host.compile_batch(["LMAEGLYN"], False, 4, 0, 1)
    kmers, progs = host.parse_blob(blob)
    assert kmers.size == 0 and status == [0]
    got = eval_program(progs[0][0], [tuple(int(x) for x in o) for o in progs[0][1]], np.zeros((0, 1), dtype=np.uint64), ones_mask(1))
    assert int(got[0]) == 1


def test_config1_through_host_compiler(host, oracle, golden):
    """BASELINE configs[0] on the CPU path: candidate bins {0,1,3} for A(C+|G+)T."""
    import os
    from conftest import GOLDEN
    files = [os.path.join(GOLDEN, "dna_example_split", "sequence%d.fa" % i) for i in range(1, 6)]
    per_bin = [sum((host.record_values(s, 3, dna=True, wraparound=True) for _, s in oracle.read_fasta(f)), []) for f in files]
    m = oracle.compute_bitcount(max(map(len, per_bin)), 0.05)
    ox = oracle.Index.ibf(5, m, 3, dna=True, k=3)
    for b, v in enumerate(per_bin):
        ox.emplace(v, b)
    blob, status, _ = host.compile_batch(["A(C+|G+)T"], True, 3, 0, 5)
    kmers, progs = host.parse_blob(blob)
    got = eval_program(progs[0][0], [tuple(int(x) for x in o) for o in progs[0][1]], ox.probe(kmers), ones_mask(5))
    assert [b for b in range(5) if (int(got[0]) >> b) & 1] == golden("config1_masks.json")["quirk"]["candidate_bins"]


def test_level_schedule_is_race_free(host, oracle):
    """Version-2 blobs: within one dependency level no op may read a slot another op of the level
    writes, plain writes are unique, and only accumulations (dst |= src) may share a destination."""
    NO = 0xFFFFFFFF
    qs = PEPTIDE_QUERIES + random_prosite_motifs(30, 21)
    blob, status, _ = host.compile_batch(qs, False, 4, 0, 256)
    _, progs = host.parse_blob(blob)
    levels = host.blob_levels(blob)
    assert levels is not None and len(levels) == len(progs)
    deep = 0
    for (n_slots, ops), ends in zip(progs, levels):
        if len(ops) == 0:
            assert ends == []
            continue
        assert ends[-1] == len(ops) and ends == sorted(ends)
        deep = max(deep, len(ends))
        begin = 0
        for end in ends:
            writes, accs, reads = {}, set(), {}
            for i in range(begin, end):
                k, d, a, b = (int(x) for x in ops[i])
                if k == NO and (d == a or d == b):
                    accs.add(d)
                    reads.setdefault(b if d == a else a, set()).add(i)
                else:
                    assert d not in writes, "two plain writes to one slot in a level"
                    writes[d] = i
                    reads.setdefault(a, set()).add(i)
                    reads.setdefault(b, set()).add(i)
            for d, i in writes.items():
                assert d not in accs
                assert reads.get(d, set()) <= {i}, "slot written and read by different ops of one level"
            for d in accs:
                assert not reads.get(d), "accumulated slot read in the same level"
            begin = end
    assert deep >= 5


def test_graphviz_dump_of_config1(host, golden):
    """`tetrex query -d`: node shapes/labels as in the reference's print_graph and exactly the arcs of
    the hand-traced config-1 graph."""
    g = golden("kgraph_config1.json")
    dot = host.kgraph_dot(g["postfix"], g["k"])
    assert dot.startswith('digraph kGraph\n{\n\trankdir="LR";\n') and dot.endswith("}")
    assert '\t0 [shape=point label=""];' in dot and '\t13 [shape=doublecircle label=""];' in dot
    assert '\t1 [label="A"];' in dot and '\t12 [label="T"];' in dot and '\t4 [label="Ø"];' in dot and '\t3 [label="•"];' in dot
    arcs = sorted(tuple(int(x) for x in line.strip(";\t").split("->")) for line in dot.splitlines() if "->" in line)
    assert arcs == sorted(map(tuple, g["arcs"]))
    # an augmented graph shows GAP nodes and hides the bypassed region
    aug = host.kgraph_dot(host.translate("LMAEG.{4}WWYN"), 4, augment=True)
    assert 'label="GAP"' in aug and aug.count("->") < host.kgraph_dot(host.translate("LMAEG.{4}WWYN"), 4).count("->")
