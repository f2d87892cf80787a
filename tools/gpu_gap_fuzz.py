#!/usr/bin/env python3
"""-a / -g on the GPU against the CPU oracle with generated motifs: sequences of the library cut into motifs whose inner
residues are turned into wildcard runs `.{n}` / `.{n,n+1}` (the gap sets -a bypasses; at most two lengths per set: a guard Split
keeps only the first and the last, in the reference in hash-set order) and classes, through libtetrex_query with -a alone and with
a device-resident d-gram index (-g, gaps 1..10); every mask against oracle.query_aug, motifs on which the reference's state-merge
quirk fires left out.  Usage on the GPU box: tools/gpu_gap_fuzz.py [motifs] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import oracle as O
from tetrex_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
AA = "ACDEFGHIKLMNPQRSTVWY"
rng = np.random.default_rng(seed)
capi.init(0)
bins = 200
seqs = ["".join(rng.choice(list(AA), size=400)) for _ in range(bins)]
ox = O.Index.ibf(bins, O.compute_bitcount(400, 0.05), 3, dna=False, k=4)
for b, s in enumerate(seqs):
    ox.emplace(O.decompose(s, 4, dna=False), b)
codes = [O.dgram_codes(s, 1, 10) for s in seqs]
dg = O.Index.ibf(bins, O.compute_bitcount(max(len(c) for c in codes), 0.05), 3, dna=False, k=4)
for b, c in enumerate(codes):
    dg.emplace(c, b)
motifs = []
while len(motifs) < n:
    s = seqs[int(rng.integers(0, bins))]
    at = int(rng.integers(0, 360))
    L = int(rng.integers(12, 30))
    w = s[at:at + L]
    out, i = [], 0
    while i < len(w):
        r = rng.random()
        if 4 <= i < len(w) - 6 and r < 0.15:  # a gap: the residues it covers are skipped in the motif
            g = int(rng.integers(2, 9))
            if i + g + 4 > len(w):
                g = max(1, len(w) - 4 - i)
            out.append(".{%d}" % g if rng.random() < 0.6 else ".{%d,%d}" % (g - 1 if g > 1 else g, g if g > 1 else g + 1))
            i += g
        elif r < 0.25:
            out.append("[" + "".join(sorted(set(w[i] + "".join(rng.choice(list(AA), size=2))))) + "]")
            i += 1
        else:
            out.append(w[i])
            i += 1
    motifs.append("".join(out))
so, sd = ox.shape(), dg.shape()
ix = capi.Index.upload_ibf(bins, so["bin_size"], 3, ox.words())
dx = capi.Index.upload_ibf(bins, sd["bin_size"], 3, dg.words())
bad = 0
for aux, mn, mx in ((None, 0, 0), (dx, 1, 10)):
    got, status, stats = ix.query_masks_gapped(motifs, False, 4, augment=True, dgram=aux, min_gap=mn, max_gap=mx)
    checked = quirks = refused = hits = 0
    for i, q in enumerate(motifs):
        try:
            want, st = ox.query_aug(q, True, dg if aux is not None else None, mn, mx)
        except Exception:  # noqa: BLE001 - the reference path cannot search it either
            if status[i] == 0:
                print("MISMATCH: %r runs here, the oracle refuses it" % q); bad += 1
            refused += 1
            continue
        if st["quirk_merges"]:
            quirks += 1
            continue
        if status[i] != 0 or not np.array_equal(got[i], want):
            print("MISMATCH (-a%s): %r status %d" % (" -g" if aux is not None else "", q, status[i])); bad += 1
        checked += 1
        hits += int(np.unpackbits(want.view(np.uint8)).sum() > 0)
    print("-a%s: %d of %d motifs compared with the oracle (%d with candidate bins; %d left out for the state-merge quirk, %d refused)" %
          (" -g" if aux is not None else "", checked, len(motifs), hits, quirks, refused), flush=True)
print("gap fuzz on the GPU: %d mismatches" % bad)
sys.exit(1 if bad else 0)
