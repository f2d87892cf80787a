"""Property-based parity (hypothesis): random regexes from the grammar the reference's lexer accepts
-> the C++ host's postfix / k-graph / compiled programs against the CPU oracle.  CPU only; the
programs are evaluated by the numpy session simulator over oracle-probed masks."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from helpers import SessionSimulator

AA = "ACDEFGHIKLMNPQRSTVWY"


def regex_strategy(alphabet, max_leaves=7):
    letter = st.sampled_from(list(alphabet))
    atom = st.one_of(
        letter,
        letter,
        st.lists(letter, min_size=2, max_size=4, unique=True).map(lambda xs: "[" + "".join(xs) + "]"),
        st.just("."),
    )

    def extend(children):
        seq = st.lists(children, min_size=2, max_size=4).map("".join)
        alt = st.lists(children, min_size=2, max_size=3).map(lambda xs: "(" + "|".join(xs) + ")")
        rep = st.tuples(children, st.sampled_from(["?", "+", "*", "{2}", "{1,2}", "{0,2}", "{2,3}"])).map(
            lambda t: ("(" + t[0] + ")" if len(t[0]) > 1 and not (t[0].startswith("[") and t[0].endswith("]") and t[0].count("[") == 1) else t[0]) + t[1])
        return st.one_of(seq, seq, alt, rep)
    return st.recursive(atom, extend, max_leaves=max_leaves)


@pytest.fixture(scope="module")
def host():
    from tetrex_amd import host as H
    H.lib()
    return H


@pytest.fixture(scope="module")
def indexes(oracle):
    rng = np.random.default_rng(7)
    pep = oracle.Index.ibf(130, 2053, 3, dna=False, k=4)
    for b in range(130):
        pep.emplace(rng.integers(0, 1 << 20, size=900, dtype=np.uint64), b)
    dna = oracle.Index.ibf(70, 257, 2, dna=True, k=3)
    for b in range(70):
        dna.emplace(rng.integers(0, 1 << 6, size=10, dtype=np.uint64), b)
    return dict(pep=pep, dna=dna)


@settings(max_examples=300, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(rx=regex_strategy(AA))
def test_translate_and_kgraph_agree_with_the_oracle(host, oracle, rx):
    postfix = oracle.translate(rx)
    assert host.translate(rx) == postfix
    assert host.preprocess(rx, False, 4, 0)[0] == oracle.trim_regex(rx)
    if not postfix:
        return
    for k in (3, 5):
        try:
            want = oracle.kgraph(postfix, k)
        except oracle.OracleError:
            with pytest.raises(host.HostError):
                host.kgraph(postfix, k)
            continue
        got = host.kgraph(postfix, k)
        assert got["labels"] == want["labels"] and got["succ"] == [tuple(s) for s in want["succ"]]


# the product's defaults, nearly / really everything dense, and the same on an executor that keeps live lists (tracked blocks)
DENSE = [dict(), dict(min_states=2, sparse_below=3), dict(min_states=1, sparse_below=1), dict(tracked=2), dict(tracked=2, min_states=2, sparse_below=3)]
EVIDENCE = [None, None, "dense", "thin"]  # what the run knows about the index: nothing (it asks: the product on a fresh index), lists saturate, states die out


def _check_masks(host, oracle, ox, rx, dna, k, budget, dense=0, evidence=None):
    """Dense DP steps are on, as in the product on a flat IBF: a generated regex such as ((.*)*)* saturates every list of
    its (large) k-graph, which enumerated state by state is millions of ops — nothing a Python test double can run."""
    import os
    sim = SessionSimulator(ox, 1)
    before = os.environ.get("TETREX_DENSE_EVIDENCE")
    if evidence:
        os.environ["TETREX_DENSE_EVIDENCE"] = evidence
    else:
        os.environ.pop("TETREX_DENSE_EVIDENCE", None)
    try:
        status, _ = host.run_staged([rx], dna, k, 0, ox.bins, sim.stage, budget, dense=DENSE[dense])
    finally:
        if before is None:
            os.environ.pop("TETREX_DENSE_EVIDENCE", None)
        else:
            os.environ["TETREX_DENSE_EVIDENCE"] = before
    try:
        want, quirks = ox.expected_mask(rx)
    except oracle.OracleError:
        assert status[0] != 0
        return
    assert status[0] == 0
    assert np.array_equal(sim.result(0), want), rx


@settings(max_examples=200, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(rx=regex_strategy(AA, max_leaves=6), budget=st.sampled_from([0, 1, 5]), dense=st.sampled_from([0, 0, 1, 2, 3, 4]), evidence=st.sampled_from(EVIDENCE))
def test_peptide_masks_equal_the_oracle(host, oracle, indexes, rx, budget, dense, evidence):
    _check_masks(host, oracle, indexes["pep"], rx, False, 4, budget, dense, evidence)


@settings(max_examples=200, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(rx=regex_strategy("ACGT", max_leaves=8), budget=st.sampled_from([0, 2]), dense=st.sampled_from([0, 1, 2, 3, 4]), evidence=st.sampled_from(EVIDENCE))
def test_dna_masks_equal_the_oracle(host, oracle, indexes, rx, budget, dense, evidence):
    _check_masks(host, oracle, indexes["dna"], rx, True, 3, budget, dense, evidence)
