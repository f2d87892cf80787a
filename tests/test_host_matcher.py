"""The verification matcher (host/matcher.hpp; the reference uses RE2: include/query.h:103,148, src/query.cpp:194-237):
leftmost-first for DNA, leftmost-longest for peptides, successive non-overlapping matches, linear time without recursion."""
import os
import re
import subprocess
import time

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def host():
    from tetrex_amd import host as H
    H.lib()
    return H


def _consume_loop(finditer_like, text):
    """The FindAndConsume loop of src/query.cpp:206-216 on top of a 'first match at or after pos' function.  Each call of
    FindAndConsume searches the REMAINING text as if it were the whole text (`^` holds at its beginning): the text is
    cut where a non-empty match ended."""
    out, pos, base = [], 0, 0
    while pos <= len(text):
        m = finditer_like(text[base:], pos - base)
        if m is None:
            break
        s, e = m[0] + base, m[1] + base
        out.append((s, e - s))
        if e > s:
            pos = base = e
        else:
            pos = s + 1
    return out


def test_leftmost_first_equals_pythons_re_on_the_readme_and_prosite_motifs(host):
    rng = np.random.default_rng(1)
    dna = "".join(rng.choice(list("ACGT"), size=5000))
    pep = "".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), size=20000))
    cases = [("A(C+|G+)T", dna), ("AC+G", dna), ("(AC)*GT", dna), ("A[^C]GT", dna), ("AC{1,3}G", dna), ("A.T", dna), ("^ACG|GT$", dna),
             ("N[^P][ST][^P]", pep), ("K[RK]{2,3}DE", pep), ("[ST].[RK]", pep), ("C.{2,4}C.{3}[LIVMFYWC]", pep), ("L(MA)+EG|W.{2}[LIVM]", pep)]
    for rx, text in cases:
        pat = re.compile("(" + rx + ")")
        want = _consume_loop(lambda t, pos: (lambda m: m.span() if m else None)(pat.search(t, pos)), text)
        assert host.regex_find_all("(" + rx + ")", text, posix=False) == want, rx
    # the reference consumes the input: what is left after a match is a text of its own, `^` holds at its beginning
    assert host.regex_find_all("(^M.K)", "MAKMAKXMAK", posix=False) == [(0, 3), (3, 3)]
    assert host.regex_find_all("(^M.K)", "MAKMAKXMAK", posix=True) == [(0, 3), (3, 3)]
    assert host.regex_find_all("(^AC|GT)", "ACACGTACGT", posix=False) == [(0, 2), (2, 2), (4, 2), (6, 2), (8, 2)]


def test_leftmost_longest_differs_from_leftmost_first_where_it_should(host):
    # POSIX: the longest match from the leftmost start; RE2 default / Perl: the first alternative that matches
    assert host.regex_find_all("(LM|LMAE)G?", "XXLMAEGXX", posix=True) == [(2, 5)]
    assert host.regex_find_all("(LM|LMAE)G?", "XXLMAEGXX", posix=False) == [(2, 2)]
    assert host.regex_find_all("(A|AC|ACG)(GT|T)?", "ACGT", posix=True) == [(0, 4)]
    assert host.regex_find_all("(A|AC|ACG)(GT|T)?", "ACGT", posix=False) == [(0, 1)]
    # anchors are text anchors
    assert host.regex_find_all("^M.[^P]{2}K$", "MAGGK", posix=True) == [(0, 5)]
    assert host.regex_find_all("^M.[^P]{2}K$", "MAGGKX", posix=True) == []
    with pytest.raises(host.HostError):
        host.regex_find_all("(AC", "ACGT", posix=True)


def test_native_differential_fuzz_against_std_regex(tmp_path):
    """tests/native/matcher_fuzz.cpp: random patterns of the supported grammar x random texts, leftmost-first against
    std::regex (ECMAScript), leftmost-longest against a brute-force oracle built from std::regex_match."""
    exe = str(tmp_path / "matcher_fuzz")
    subprocess.run(["g++", "-O2", "-std=c++20", "-o", exe, os.path.join(ROOT, "tests", "native", "matcher_fuzz.cpp"),
                    os.path.join(ROOT, "tetrex_amd", "csrc", "host", "matcher.cpp")], check=True, timeout=600)
    for seed in ("11", "12"):
        r = subprocess.run([exe, "fuzz", seed, "500"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-3000:]


def test_megabyte_records_with_quantifier_motifs_take_linear_time(host):
    """What a backtracking matcher cannot do: chromosome-length records with * + {m,n} motifs — no stack
    growth, no exponential blow-up.  4 MB of sequence per motif in well under a second each."""
    rng = np.random.default_rng(2)
    dna = "".join(rng.choice(list("ACGT"), size=4_000_000))
    t0 = time.perf_counter()
    hits = host.regex_find_all("(A(C+|G+)T)", dna, posix=False)
    assert len(hits) > 10000 and all(dna[s] == "A" and dna[s + n - 1] == "T" for s, n in hits[:200])
    assert host.regex_find_all("((AC)*GTTTTTTTTTTTT)", dna, posix=False) == [(m.start(), m.end() - m.start()) for m in re.finditer("(AC)*GTTTTTTTTTTTT", dna)]
    # the classic exponential case for backtrackers: nested quantifiers over a long run that then fails to match
    run = "A" * 2_000_000 + "C"
    assert host.regex_find_all("((A+)+G)", run, posix=True) == []
    assert host.regex_find_all("((A+)+C)", run, posix=True) == [(0, 2_000_001)]
    assert host.regex_find_all("((A|AA)+C)", run, posix=False) == [(0, 2_000_001)]
    assert time.perf_counter() - t0 < 20


def test_the_required_literal_is_contained_in_every_match(host):
    """The prefilter in front of the automata (Matcher::may_match): the longest run of plain bytes on the pattern's spine.  It must
    be a string EVERY match contains — checked against Python's re on random texts — and the known cases come out as expected."""
    known = {"(LMA(E|Q)GLYN)": "GLYN", "(C.{2,4}C.{3}[LIVMFYWC].{8}H.{3,5}H)": "C", "(AB|CD)": "", "(A(BC)+DE)": "BC", "((AB)*CDE)": "CDE",
             "(N[^P][ST][^P])": "N", "(LMA{3}E)": "LM", "(K[RK]{2,3}DE)": "DE", "(^MAEG$)": "MAEG", "(L(MA)*EG)": "EG", "(AC?GT)": "GT", "(.*LMAE.+)": "LMAE",
             "(W..[LIVM]D)": "W", "(A{2,4}C)": "A"}
    for rx, lit in known.items():
        assert host.regex_required_literal(rx) == lit, rx
    rng = np.random.default_rng(3)
    alphabet = list("ACDE")
    pieces = ["A", "C", "D", "E", ".", "[AC]", "[^D]", "(A|CD)", "(DE)+", "C?", "A*", "E{2}", "D{1,2}", "(AC)?", "(C|D)+"]
    for _ in range(300):
        rx = "(" + "".join(rng.choice(pieces, size=int(rng.integers(2, 7)))) + ")"
        lit = host.regex_required_literal(rx)
        pat = re.compile(rx)
        for _ in range(20):
            text = "".join(rng.choice(alphabet, size=40))
            for m in pat.finditer(text):
                assert lit in m.group(0), (rx, lit, m.group(0))
            # and find_all is unchanged by the prefilter: the same matches as Python's leftmost-first search loop
            want = _consume_loop(lambda t, pos: (lambda m: m.span() if m else None)(pat.search(t, pos)), text)
            assert host.regex_find_all(rx, text, posix=False) == want, (rx, text)
