"""CPU tests of the dense DP steps (include/txq_program.h version 3; host/compiler.cpp densify / dense_step /
materialise, level scheduling with block hazards in schedule_levels_into).  The device session is replaced by
the numpy simulator (helpers.SessionSimulator), which evaluates DENSE_ZERO / STEP / REDUCE from their
definition over oracle-probed masks; the final masks must equal the oracle's collect()
(reference include/otf_collector.h:341-393) whatever the thresholds that decide where lists go dense."""
import numpy as np
import pytest

from helpers import SessionSimulator
from motifs import PEPTIDE_QUERIES, DNA_QUERIES, random_prosite_motifs


@pytest.fixture(autouse=True)
def _lists_saturate(monkeypatch):
    """These tests are about the dense machinery, on small (sparse) indexes: tell the expansion that lists saturate instead
    of letting it find out that they do not (tests of that protocol set TETREX_DENSE_EVIDENCE themselves)."""
    monkeypatch.setenv("TETREX_DENSE_EVIDENCE", "dense")


@pytest.fixture(scope="module")
def host():
    from tetrex_amd import host as H
    H.lib()
    return H


def _index(oracle, bins, m, h, k, dna, per_bin, seed, reduction=0):
    ox = oracle.Index.ibf(bins, m, h, dna=dna, k=k, reduction=reduction)
    rng = np.random.default_rng(seed)
    bits = (2 if dna else 5) * k
    for b in range(bins):
        ox.emplace(rng.integers(0, 1 << min(bits, 62), size=per_bin, dtype=np.uint64), b)
    return ox


def _run(host, ox, queries, dna, k, dense, per_query=0, gaps=None, reduction=0, augment=False, wants=None):
    """wants: a dict that keeps the oracle's answers between calls on the same index contents (the oracle enumerates every state)."""
    sim = SessionSimulator(ox, len(queries))
    status, stats = host.run_staged(queries, dna, k, reduction, ox.bins, sim.stage, per_query, 0, gaps=gaps, dense=dense)
    checked = 0
    for i, q in enumerate(queries):
        try:
            if wants is not None and q in wants:
                if wants[q] is None:
                    raise ValueError(q)
                want, quirks = wants[q]
            else:
                try:
                    want, quirks = ox.expected_mask(q, augment=augment)
                except Exception:
                    if wants is not None:
                        wants[q] = None
                    raise
                if wants is not None:
                    wants[q] = (want, quirks)
        except Exception:
            assert status[i] != 0
            continue
        assert status[i] == 0, q
        assert np.array_equal(sim.result(i), want), q
        checked += 1
    return checked, stats, sim


THRESHOLDS = [dict(min_states=1, sparse_below=1), dict(min_states=2, sparse_below=3), dict(min_states=20, sparse_below=9), dict()]


@pytest.mark.parametrize("dense", THRESHOLDS, ids=["everything", "2/2", "20/8", "defaults"])
def test_peptide_queries_with_dense_steps(host, oracle, dense):
    ox = _index(oracle, bins=200, m=4099, h=3, k=4, dna=False, per_bin=1500, seed=1)
    qs = [q for q in PEPTIDE_QUERIES if "{2,4}C" not in q] + random_prosite_motifs(25, 3, wildcard=0.1, ranges=0.05)
    checked, stats, sim = _run(host, ox, qs, False, 4, dense)
    assert checked >= len(qs) - 8
    assert sim.dense_steps > 100 and sim.dense_kinds[0] > 0
    if dense.get("sparse_below") == 1:
        assert sim.dense_kinds[2] > 10  # blocks reach the Match node: DENSE_REDUCE into RESULT


@pytest.mark.parametrize("per_query", [1, 7, 64])
def test_blocks_persist_across_stages(host, oracle, per_query):
    """A stage budget of a few ops cuts the queries between dense steps: blocks written in one stage are read,
    zeroed and recycled in later ones."""
    ox = _index(oracle, bins=130, m=2053, h=3, k=4, dna=False, per_bin=800, seed=2)
    # three literal residues first: the oracle's result is then well defined (no quirk merges)
    qs = ["LMK.{1,3}A[DE]..GK", "WKL..[LIVM]D.[FY]", "LMKA.C.E.GH", "KRK[RK]{2,3}.DE", "CLM.{2,4}C...[LIVMFYWC]"]
    checked, stats, sim = _run(host, ox, qs, False, 4, dict(min_states=4, sparse_below=3), per_query=per_query)
    assert checked == len(qs) and sim.dense_steps > 20 and (stats["stages"] >= 2 or per_query > 7)


def test_dna_dense_steps_probe_canonical_kmers(host, oracle):
    ox = _index(oracle, bins=70, m=257, h=3, k=3, dna=True, per_bin=8, seed=3)
    checked, stats, sim = _run(host, ox, DNA_QUERIES + ["A..T.G", "AC.{1,3}GT", "[AC]..[GT]A"], True, 3, dict(min_states=1, sparse_below=1))
    assert checked >= 12 and sim.dense_steps > 20
    ox = _index(oracle, bins=64, m=1021, h=2, k=5, dna=True, per_bin=60, seed=4)
    checked, stats, sim = _run(host, ox, ["ACG..T.GA", "A.{2,4}CGT.A", "GATTACA", "AC[GT]..[AC]CGT"], True, 5, dict(min_states=3, sparse_below=3))
    assert checked >= 3 and sim.dense_steps > 5


def test_reduced_alphabet_k5(host, oracle):
    ox = _index(oracle, bins=96, m=2053, h=2, k=5, dna=False, per_bin=500, seed=5, reduction=1)
    qs = ["LMA..E[DE]GLY", "WK.{1,2}[LIVM]D.F", "AC.DE.GH", "M[KR]..S[ST].L", "LMAEGLYN"]
    checked, stats, sim = _run(host, ox, qs, False, 5, dict(min_states=2, sparse_below=3), reduction=1)
    assert checked >= 4 and sim.dense_steps > 5


def test_no_block_left_falls_back_to_enumerated_states(host, oracle):
    """With room for a single block (or a pool too small for even one) the steps out of a block have nowhere to
    accumulate: the block is enumerated again and the query goes on with ordinary ops — same masks."""
    ox = _index(oracle, bins=100, m=2053, h=3, k=4, dna=False, per_bin=700, seed=6)
    qs = ["LMK.{1,3}A[DE]..GK", "WKL..[LIVM]D.[FY]", "LMKA.C.E.GH"]
    for dense in (dict(min_states=2, sparse_below=1, max_blocks=1), dict(min_states=2, sparse_below=1, max_blocks=2),
                  dict(min_states=2, sparse_below=1, pool_bytes=1)):
        checked, stats, sim = _run(host, ox, qs, False, 4, dense)
        assert checked == len(qs)


def test_gap_nodes_reduce_a_block(host, oracle):
    """-a without a d-gram index: a state crossing a Gap node restarts its k-mer, so a whole block collapses
    into one state (DENSE_REDUCE into an ordinary slot)."""
    ox = _index(oracle, bins=100, m=2053, h=3, k=4, dna=False, per_bin=700, seed=7)
    # residue classes (too few paths to be bypassed) make the list dense, the wildcard run behind them becomes Gap nodes
    qs = ["LMK[DE][KR][ST][LIV].{3,6}FK", "WKL[LIVM][DE][FY].{2,4}GH[KR]K", "CLMAC[DE][ST].{2,3}GH[LIV]K"]
    gaps = dict(augment=1)
    checked, stats, sim = _run(host, ox, qs, False, 4, dict(min_states=2, sparse_below=1), gaps=gaps, augment=True)
    assert checked >= 2 and sim.dense_steps >= 6 and sim.dense_kinds[2] >= 2


def test_dense_cuts_the_host_work_of_a_saturated_motif(host, oracle):
    """The point of it all: a wildcard run saturates the state list (20^3 states at k = 4); enumerated, every further
    residue class costs an op per state and residue; as a block it costs one op."""
    ox = oracle.Index.ibf(64, 257, 3, dna=False, k=4)
    ox.set_words(np.full(257, np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64))
    qs = ["LMA.{3,5}[DE]..[LIVM]GK.H"]
    _, plain, _ = _run(host, ox, qs, False, 4, None)
    _, dense, sim = _run(host, ox, qs, False, 4, dict())
    assert plain["ops"] > 200000 and dense["ops"] < plain["ops"] / 50
    assert int(sim.result(0)[0]) == 0xFFFFFFFFFFFFFFFF


_NESTED_WANTS = {}  # the two runs below use one index (same seed)


@pytest.mark.parametrize("dense", [dict(), dict(min_states=1, sparse_below=1)], ids=["defaults", "everything"])
def test_nested_stars_that_enumerated_state_by_state_would_exceed_any_op_budget(host, oracle, dense):
    """((.*)*)* and friends (the first was found by the fuzz test): every list of a large k-graph is saturated — millions
    of states for the oracle.  As dense blocks the queries need a few thousand ops; the product used to give up on them
    at 8 M ops.  The starred ones make the reference merge states of different length (quirk merges: its result is
    implementation-defined there), so parity is asserted on the '+' forms and completion on all."""
    ox = _index(oracle, bins=96, m=2053, h=3, k=4, dna=False, per_bin=700, seed=11)
    qs = ["LMK(.+)+HKD", "LMK(.+)+(.+)+HKD", "LMK((.+)+)+KDE", "LMK(.+)+D(.+)+HK", "((.*)*)*", "LMK((.*)*)*KDE", "(.+)+LMK(.*)*"]
    checked, stats, sim = _run(host, ox, qs, False, 4, dense, wants=_NESTED_WANTS)
    assert checked >= 4 and stats["ops"] < 5_000_000


def test_queries_begin_in_waves_when_block_memory_is_short(host, oracle):
    """A pool of a few blocks for a dozen wildcard motifs: the ones that have not begun wait while those under way finish
    and hand their blocks back, instead of all starting at once and most falling back to enumerated states.  Same masks,
    several stages, about the ops of an unlimited pool."""
    ox = _index(oracle, bins=100, m=2053, h=3, k=4, dna=False, per_bin=700, seed=12)
    qs = ["LMK.{1,3}A[DE]..GK", "WKL..[LIVM]D.[FY]", "LMKA.C.E.GH", "CLM.{2,4}C...[LIVMFYWC]", "KRK..[DE].GH", "HKL.{1,2}[ST]..P",
          "LMK..A..GK", "WKL.[LIVM]..D.[FY]", "LMKA..E.GH.", "CLM..C...[LIVMFYWC]", "KRK.[DE]..GH", "HKL..[ST]..P"]
    block = 21 ** 3 * 128
    checked, free, _ = _run(host, ox, qs, False, 4, dict(slot_bytes=128))
    checked2, tight, sim = _run(host, ox, qs, False, 4, dict(slot_bytes=128, pool_bytes=12 * block))
    assert checked == checked2 == len(qs)
    # (a pool that is running short keeps its blocks for the long lists — the thresholds of round 2, DenseOptions::short_min_states —
    # so the yardstick is the unlimited run at those thresholds; the unlimited run at the product's makes blocks of short lists, too)
    _, roomy32, _ = _run(host, ox, qs, False, 4, dict(slot_bytes=128, min_states=32, sparse_below=17))
    assert free["ops"] <= roomy32["ops"]
    assert tight["stages"] > free["stages"] and tight["ops"] < 3 * roomy32["ops"]


def test_a_block_that_waits_for_the_answer_is_cleared_whole(host, oracle, monkeypatch):
    """Found by tests/test_fuzz_parity.py with the thresholds of round 3 (`D?[AC]+.?`, k = 4): a list that already owns a block
    pauses to ask how states fare on the index (TETREX_DENSE_EVIDENCE=ask) AFTER its block's ZERO has been cut down to the shape
    seen so far; in the next stage the list's own states join the block with a residue that shape does not have, and the ZERO —
    shipped with the first stage — had not cleared those entries.  The pause now widens the ZERO to the whole block again."""
    monkeypatch.setenv("TETREX_DENSE_EVIDENCE", "ask")
    rng = np.random.default_rng(7)
    ox = oracle.Index.ibf(130, 2053, 3, dna=False, k=4)
    for b in range(130):
        ox.emplace(rng.integers(0, 1 << 20, size=900, dtype=np.uint64), b)
    qs = ["D?[AC]+.?", "E?[KR]+.?", "D?[ACD]+..?", "[DE]?[AC]+.?G"]
    for dense in (dict(), dict(min_states=8, sparse_below=5), dict(min_states=4, sparse_below=3)):
        checked, stats, sim = _run(host, ox, qs, False, 4, dense)
        assert checked == len(qs) and sim.dense_steps > 0


@pytest.mark.parametrize("kind", ["saturated", "sparse"])
def test_the_expansion_asks_before_its_first_block(host, oracle, monkeypatch, kind):
    """Nothing known about the index (TETREX_DENSE_EVIDENCE=ask, the product's default for a fresh index): a query pauses
    before the first list that could become a block and reads the fill of its probed states' masks from the answers
    (1 + floor(log2(bits)), as txq_session_stage gives them).  Saturated index (every k-mer in most bins): blocks from then
    on; sparse index: no block at all, states are enumerated and pruned.  The masks are the oracle's either way."""
    monkeypatch.setenv("TETREX_DENSE_EVIDENCE", "ask")
    bins, k = 256, 3
    if kind == "saturated":
        ox = oracle.Index.ibf(bins, 4099, 2, dna=False, k=k)
        every = np.arange(1 << 15, dtype=np.uint64)  # all 3-mers over 5-bit codes
        for b in range(bins):
            if b % 5:
                ox.emplace(every, b)
    else:
        ox = _index(oracle, bins=bins, m=4099, h=2, k=k, dna=False, per_bin=300, seed=31)
    qs = ["LMKA..CDE.GH", "WKLA.{1,3}CDEF", "ACDEF...GHIKL", "LMKACDE"]
    checked, stats, sim = _run(host, ox, qs, False, k, dict(min_states=8, sparse_below=4))
    assert checked == len(qs)
    if kind == "saturated":
        assert sim.dense_steps > 0 and stats["stages"] >= 2
    else:
        assert sim.dense_steps == 0


@pytest.mark.parametrize("dense", [dict(min_states=1, sparse_below=1, tracked=2), dict(min_states=8, sparse_below=4, tracked=2), dict(tracked=2)],
                         ids=["everything", "8/3", "defaults"])
def test_tracked_blocks_keep_the_meaning_of_the_dense_ops(host, oracle, dense):
    """Tracked (sparse) blocks (include/txq_program.h): the program says that its blocks carry live lists and every one of its
    dense ops repeats it; a list becomes a block whatever its shape.  The ops mean what they meant, so the simulator — which
    evaluates them from their definition — must arrive at the oracle's masks."""
    ox = _index(oracle, bins=200, m=4099, h=3, k=4, dna=False, per_bin=1500, seed=1)
    qs = [q for q in PEPTIDE_QUERIES if "{2,4}C" not in q] + random_prosite_motifs(25, 3, wildcard=0.1, ranges=0.05)
    checked, stats, sim = _run(host, ox, qs, False, 4, dense)
    assert checked >= len(qs) - 8
    assert sim.dense_steps > 100 and sim.tracked_ops > 100 and any(sim.tracked)


def test_a_run_of_unprobed_wildcard_states_is_one_fill(host, oracle):
    """`LM...` at k = 6: the 21^3 states behind the wildcards have not been probed when they reach full length — they all
    carry ONES and fill their shape exactly, so the list becomes a block with one DENSE_FILL instead of 9261 scatter ops."""
    ox = _index(oracle, bins=64, m=2053, h=2, k=6, dna=False, per_bin=400, seed=21)
    qs = ["LM...KDE", "WK....[DE]H", "LMK.{2,3}A..E"]
    for dense in (dict(min_states=32, sparse_below=4, tracked=2), dict(min_states=32, sparse_below=4)):
        checked, stats, sim = _run(host, ox, qs, False, 6, dense)
        assert checked == len(qs)
        assert sim.dense_kinds[3] >= 3 and stats["ops"] < 40000


def test_sparse_evidence_turns_queries_to_tracked_blocks(host, oracle, monkeypatch):
    """TETREX_DENSE_EVIDENCE=thin (what a run learns on an index where states die out) and an executor that keeps live
    lists: lists become tracked blocks regardless of their shape; the same run on an executor without lists enumerates."""
    monkeypatch.setenv("TETREX_DENSE_EVIDENCE", "thin")
    ox = _index(oracle, bins=128, m=4099, h=2, k=5, dna=False, per_bin=600, seed=22)
    qs = ["LMK..A[DE]..GK", "WKL.[LIVM]D..[FY]", "CLM.{2,4}C...[LIVMFYWC]"]
    checked, with_lists, sim = _run(host, ox, qs, False, 5, dict(tracked=1))
    assert checked == len(qs) and any(sim.tracked) and sim.dense_steps > 5
    checked, without, sim2 = _run(host, ox, qs, False, 5, dict())
    assert checked == len(qs) and not any(sim2.tracked)
    assert with_lists["ops"] <= without["ops"]


def test_tracked_programs_keep_the_states_of_their_first_residues_as_blocks_too(host, oracle):
    """`C....C...` at k = 6: 20^4 states behind the four wildcards before the first k-mer is complete.  A tracked program keeps
    such lists as blocks as well — one block per state length, rolled forward by TXQ_DENSE_NOPROBE steps (no k-mer to look
    up yet) — instead of enumerating them on the host.  Masks = the oracle's; the host's state count collapses."""
    ox = _index(oracle, bins=64, m=4099, h=2, k=6, dna=False, per_bin=500, seed=23)
    qs = ["LM...KDE", "K.{2,3}C..[LIVM]", "L..M..K"]
    checked, plain, sim0 = _run(host, ox, qs, False, 6, dict(min_states=32, sparse_below=4))
    checked2, tracked, sim = _run(host, ox, qs, False, 6, dict(min_states=32, sparse_below=4, tracked=2))
    assert checked == checked2 == len(qs)
    assert sim.noprobe_steps >= 4 and getattr(sim0, "noprobe_steps", 0) == 0
    assert tracked["states"] * 10 < plain["states"]
