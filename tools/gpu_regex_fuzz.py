#!/usr/bin/env python3
"""Random regexes (the grammar of tests/test_fuzz_parity.py, drawn with hypothesis' generators from a seed) through the GPU path
in batches, every mask against the CPU oracle: the device reads whatever the arena holds where a ZERO has not cleared, which the
numpy simulator of the CPU tests only sees as a poisoned entry.  Peptide k = 4 on 130 bins and DNA k = 3 on 70 bins (the fuzz
tests' indexes) and peptide k = 4 on 1024 bins (16-byte lanes), product defaults (a fresh index asks how states fare on it), then TETREX_DENSE_EVIDENCE=dense / thin.
Usage on the GPU box: tools/gpu_regex_fuzz.py [regexes per index] [seed]"""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
warnings.filterwarnings("ignore")
import hypothesis
from hypothesis import strategies as st
import oracle as O
from tetrex_amd import capi
from test_fuzz_parity import regex_strategy, AA

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
capi.init(0)


def draw(strategy, count, seed):
    out = []

    @hypothesis.seed(seed)
    @hypothesis.settings(max_examples=count, database=None, deadline=None, suppress_health_check=list(hypothesis.HealthCheck), phases=[hypothesis.Phase.generate])
    @hypothesis.given(strategy)
    def collect(rx):
        out.append(rx)
    collect()
    return list(dict.fromkeys(out))


rng = np.random.default_rng(7)
pep = O.Index.ibf(130, 2053, 3, dna=False, k=4)
for b in range(130):
    pep.emplace(rng.integers(0, 1 << 20, size=900, dtype=np.uint64), b)
dna = O.Index.ibf(70, 257, 2, dna=True, k=3)
for b in range(70):
    dna.emplace(rng.integers(0, 1 << 6, size=10, dtype=np.uint64), b)
wide = O.Index.ibf(1024, 8191, 3, dna=False, k=4)  # 16-word masks: the 16-byte lanes of the step kernels
for b in range(1024):
    wide.emplace(rng.integers(0, 1 << 20, size=2500, dtype=np.uint64), b)
# HIBFs: a random irregular 3-level tree, a tree as a layout algorithm shapes it, a regular 16 x 64 tree (tests/helpers.py)
from helpers import layout_hibf, random_hibf, regular_hibf
trees = {}
tree_seed = int(os.environ.get("FUZZ_TREE_SEED", "0"))  # other shapes of the two general trees (0: the ones of the committed runs)
ox_t, descs, _ = random_hibf(O, 21 + tree_seed, user_bins=300, levels=3 if tree_seed % 2 == 0 else 4, n_values=60)
trees["hibf-irregular-300"] = (ox_t, descs, 300)
ox_t, descs, _ = layout_hibf(O, 5 + tree_seed, user_bins=900, tmax=32 if tree_seed % 3 == 0 else (16 if tree_seed % 3 == 1 else 64), n_values=30)
trees["hibf-layout-900"] = (ox_t, descs, 900)
ox_t, descs, _ = regular_hibf(O, 1024, 16, 200, lambda b: np.random.default_rng(b).integers(0, 1 << 20, size=200, dtype=np.uint64), h=2)
trees["hibf-regular-16x64"] = (ox_t, descs, 1024)
bad = 0
# reduced alphabets (the reduced k-graph builder: symbols buffered, equal reduced letters of a union collapse): Murphy k = 5, Li k = 4
murphy = O.Index.ibf(130, 2053, 3, dna=False, k=5, reduction=1)
li = O.Index.ibf(130, 2053, 3, dna=False, k=4, reduction=2)
for b in range(130):
    murphy.emplace(rng.integers(0, 1 << 20, size=900, dtype=np.uint64), b)
    li.emplace(rng.integers(0, 1 << 16, size=900, dtype=np.uint64), b)
cases = [("peptide", pep, False, 4, AA, 6, 0), ("dna", dna, True, 3, "ACGT", 8, 0), ("peptide-1024-bins", wide, False, 4, AA, 6, 0),
         ("peptide-murphy", murphy, False, 5, AA, 6, 1), ("peptide-li", li, False, 4, AA, 6, 2)]
cases += [(name, t[0], False, 4, AA, 6, 0) for name, t in trees.items()]
more = int(os.environ.get("FUZZ_MORE_LEAVES", "0"))  # larger regexes (the oracle enumerates every state: keep the count down)
n_shards = int(os.environ.get("FUZZ_SHARDS", "1"))        # > 1: the HIBF cases as that many sub-tree shards (column shards for the regular tree)
only = os.environ.get("FUZZ_ONLY")                    # a comma-separated choice of the cases above
for name, ox, is_dna, k, alphabet, leaves, reduction in cases:
    if only and name not in only.split(","):
        continue
    qs = draw(regex_strategy(alphabet, max_leaves=leaves + more), n, seed)
    wants = []
    for q in qs:
        try:
            wants.append(ox.expected_mask(q)[0])
        except Exception:  # noqa: BLE001 - a regex the reference path cannot search either
            wants.append(None)
    sh = ox.shape() if name not in trees else None
    for ev in (None, "dense", "thin"):
        if ev:
            os.environ["TETREX_DENSE_EVIDENCE"] = ev
        else:
            os.environ.pop("TETREX_DENSE_EVIDENCE", None)
        for chunk in (len(qs), 40, 7):  # one batch, and batches small enough to leave the table of all k-mers' masks alone
            shards = None
            if name in trees and n_shards > 1:  # FUZZ_SHARDS: the tree as sub-tree shards on this GPU, masks ORed (txq_index_upload_subtrees)
                shards = [capi.Index.upload_hibf(trees[name][2], trees[name][1], shard_rank=r, n_shards=n_shards, subtrees=True) for r in range(n_shards)]
                ix = None
            elif name in trees:
                ix = capi.Index.upload_hibf(trees[name][2], trees[name][1])
            elif n_shards > 1:  # a flat IBF as column shards, one expansion for all of them (txe_query_masks_sharded)
                shards = [capi.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words(), shard_rank=r, n_shards=n_shards) for r in range(n_shards)]
                ix = None
            else:
                ix = capi.Index.upload_ibf(ox.bins, sh["bin_size"], sh["hash_funs"], ox.words())  # a fresh index: it is asked
            checked = 0
            for at in range(0, len(qs), chunk):
                part = qs[at:at + chunk]
                got, status, stats = capi.query_masks_sharded(shards, part, is_dna, k, reduction) if shards else ix.query_masks(part, is_dna, k, reduction)
                for q, g, w, s_ in zip(part, got, wants[at:at + chunk], status):
                    if w is None:
                        if s_ == 0:
                            print("MISMATCH: %r runs here, the oracle refuses it" % q); bad += 1
                        continue
                    if s_ != 0 or not np.array_equal(g, w):
                        print("MISMATCH %s evidence %s batch %d: %r status %d" % (name, ev, chunk, q, s_)); bad += 1
                    checked += 1
            for x in (shards or [ix]):
                x.free()
            print("%s, evidence %s, batches of %d: %d of %d regexes compared with the oracle" % (name, ev or "asked", chunk, checked, len(qs)), flush=True)
print("regex fuzz on the GPU: %d mismatches" % bad)
sys.exit(1 if bad else 0)
