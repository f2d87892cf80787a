"""oracle/check_masks.py (test infrastructure): the child process bench.py's parity legs start to compare MANY masks with the CPU
oracle on all host threads, with a deadline.  Here its own bookkeeping is checked on the CPU: masks the oracle itself computed are
all accepted, a flipped bit ends the bench, and a deadline that has passed stops the child without an answer being lost."""
import numpy as np
import pytest

from helpers import random_hibf
from motifs import PEPTIDE_QUERIES


def _bench():
    import bench
    return bench


def test_flat_index_masks_are_checked_in_a_child_process(oracle):
    bench = _bench()
    bins, m, h, k = 200, 2053, 3, 4
    ox = oracle.Index.ibf(bins, m, h, dna=False, k=k)
    rng = np.random.default_rng(3)
    for b in range(bins):
        ox.emplace(rng.integers(0, 1 << 20, size=600, dtype=np.uint64), b)
    motifs = [q for q in PEPTIDE_QUERIES if "{2,4}" not in q][:20] + ["A|"]  # (the last one: a motif the reference path refuses)
    masks = np.zeros((len(motifs), ox.words_per_mask), dtype=np.uint64)
    for i, q in enumerate(motifs[:-1]):
        masks[i] = ox.expected_mask(q)[0]
    meta = {"kind": "ibf", "bins": bins, "rows": m, "h": h, "dna": False, "k": k}
    res = bench.oracle_check_rest({"words": ox.words()}, meta, motifs, masks, range(len(motifs)), 120.0, "test")
    assert res["masks_compared"] == len(motifs) - 1 and res["refused_by_the_oracle"] == 1 and res["unfinished"] == 0
    assert bench.oracle_check_rest({"words": ox.words()}, meta, motifs, masks, [], 1.0, "test")["masks_compared"] == 0
    masks[3, 0] ^= np.uint64(1 << 7)
    with pytest.raises(SystemExit) as e:
        bench.oracle_check_rest({"words": ox.words()}, meta, motifs, masks, range(len(motifs)), 120.0, "test")
    assert repr(motifs[3]) in str(e.value)
    # a deadline that is over before the child has loaded anything: stopped, nothing compared, nothing claimed
    res = bench.oracle_check_rest({"words": ox.words()}, meta, motifs[:5], masks[:5], range(3), 0.0, "test")
    assert res["stopped_at_the_deadline"] is True and res["masks_compared"] + res["unfinished"] + res["refused_by_the_oracle"] == 3


def test_hibf_masks_are_checked_in_a_child_process(oracle):
    bench = _bench()
    ox, descs, values = random_hibf(oracle, 5, user_bins=150, levels=3, n_values=30)
    motifs = ["LMA(E|Q)GLYN", "A.CD", "K[RK]DE", "W..[LIVM]D"]
    masks = np.stack([ox.expected_mask(q)[0] for q in motifs])
    res = bench.oracle_check_rest(descs, {"kind": "hibf", "bins": 150, "dna": False, "k": 4}, motifs, masks, range(len(motifs)), 120.0, "test")
    assert res["masks_compared"] == len(motifs) and res["unfinished"] == 0
