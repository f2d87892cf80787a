#!/usr/bin/env python3
"""Regenerates the golden JSON files of this directory that CAN be regenerated — from the formulas of SURVEY.md §8(c),
by an independent pure-Python model (big integers; none of the product's or the oracle's code), on the reference-held
data files that sit next to it:

  hash_kat.json       rows of the reference-built fixture ibf_idx.ibf (decoded from its legacy container), the per-k-mer
                      rows of its ten canonical 3-mers, derived-only KATs for all five hash seeds, compute_bitcount values
  config1_masks.json  BASELINE configs[0]: the 5-bin toy library (dna_example_split) as an IBF, k = 3, with and without the
                      reference's wrap-around quirk; per-k-mer masks and candidate bins of A(C+|G+)T; AC+G on the fixture
  config2_kmers.json  the packed k-mers LMA(E|Q)GLYN probes at k = 4

  python tests/golden/regenerate.py            rewrite the three files
  python tests/golden/regenerate.py --check    compare with the committed files, exit 1 on any difference

FROZEN, not regenerated (see README.md): translate.json, encoders.json (captured during the survey from pieces of the
reference compiled against throw-away stub headers — not reproducible under the current rules), kgraph_config1.json (a hand trace)."""
import json
import math
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SEEDS = [13572355802537770549, 13043817825332782213, 10650232656628343401, 16499269484942379435, 4893150838803335377]
GOLDEN = 11400714819323198485
M64 = (1 << 64) - 1


def hash_row(value, seed, bin_size):
    """seqan::hibf hash_and_fit / bin_size: seed multiply, xor-shift by countl_zero(bin_size),
    golden-ratio multiply, 128-bit fastrange."""
    shift = 64 - bin_size.bit_length()
    h = (value * seed) & M64
    h ^= h >> shift
    h = (h * GOLDEN) & M64
    return (h * bin_size) >> 64


def compute_bitcount(n, fpr):
    """IBFIndex::compute_bitcount (include/index_ibf.h:133-139): the logarithm of the false-positive rate in float."""
    return int(math.ceil(-float(n) * float(np.log(np.float32(fpr))) / (math.log(2.0) ** 2)))


CODE = {"A": 0, "C": 1, "T": 2, "G": 3}  # (c >> 1) & 3, include/nucleotide_decomposer.h:86-87


def canonical(fwd, k):
    rc, f = 0, fwd
    for _ in range(k):
        rc = (rc << 2) | ((f & 3) ^ 2)
        f >>= 2
    return min(fwd, rc)


def dna_values(seq, k, quirk):
    """decompose_record (include/nucleotide_decomposer.h:99-110); quirk: the first k symbols are rolled in a second time."""
    mask = (1 << (2 * k)) - 1
    fwd = 0
    for c in seq[:k]:
        fwd = (fwd << 2) | CODE[c]
    out = [canonical(fwd, k)]
    for c in (seq if quirk else seq[k:]):
        fwd = ((fwd << 2) & mask) | CODE[c]
        out.append(canonical(fwd, k))
    return out


def pack_dna(s):
    v = 0
    for c in s:
        v = (v << 2) | CODE[c]
    return v


def read_fasta(path):
    return "".join(line.strip() for line in open(path) if not line.startswith(">"))


def hash_kat():
    blob = open(os.path.join(HERE, "ibf_idx.ibf"), "rb").read()
    words = struct.unpack_from("<64Q", blob, 0x4E)  # legacy container: 64 rows x 1 word at offset 0x4e
    rows = lambda b: [r for r in range(64) if (words[r] >> b) & 1]
    kmers = [1, 5, 23, 7, 15, 3, 41, 37, 20, 16]  # AAC ACC CCG ACG AGG AAG TTC TCC CCA CAA, canonical values
    fixture = {"bin_size": 64, "h": 3, "k": 3, "rows_bin0": rows(0), "rows_bin1": rows(1),
               "kmer_rows": {str(v): [hash_row(v, SEEDS[i], 64) for i in range(3)] for v in kmers}}
    # the fixture's bits are exactly the union of its records' k-mer rows (this is what pins hash + layout)
    for b, name in enumerate(("file1.fa", "file2.fa")):
        got = set()
        for line in open(os.path.join(HERE, name)):
            if not line.startswith(">"):
                for v in dna_values(line.strip(), 3, quirk=False):
                    got |= {hash_row(v, SEEDS[i], 64) for i in range(3)}
        assert sorted(got) == rows(b), name
    derived = [{"value": v, "bin_size": m, "rows": [hash_row(v, s, m) for s in SEEDS]} for v, m in ((12345, 1000), (0xFFFFF, 8191), (123456789, 62500000))]
    bitcount = [{"n": n, "fpr": 0.05, "m": compute_bitcount(n, 0.05)} for n in (200000, 17, 14)]
    return {"fixture": fixture, "derived_only": derived, "bitcount": bitcount}, words


def config1(fixture_words):
    seqs = [read_fasta(os.path.join(HERE, "dna_example_split", "sequence%d.fa" % i)) for i in range(1, 6)]
    out = {"k": 3, "h": 3, "fpr": 0.05, "bins": 5}
    for name, quirk in (("quirk", True), ("plain", False)):
        per_bin = [dna_values(s, 3, quirk) for s in seqs]
        n_max = max(len(v) for v in per_bin)
        m = compute_bitcount(n_max, 0.05)
        rows_of = [set() for _ in range(5)]
        for b, vals in enumerate(per_bin):
            for v in vals:
                rows_of[b] |= {hash_row(v, SEEDS[i], m) for i in range(3)}
        mask = lambda kmer: [all(hash_row(canonical(pack_dna(kmer), 3), SEEDS[i], m) in rows_of[b] for i in range(3)) for b in range(5)]
        bits = lambda kmer: "".join("1" if x else "0" for x in reversed(mask(kmer)))  # bit string = bins 4..0
        AND = lambda *ms: [all(t) for t in zip(*ms)]
        OR = lambda *ms: [any(t) for t in zip(*ms)]
        cand = OR(mask("ACT"), AND(mask("ACC"), mask("CCT")), mask("AGT"), AND(mask("AGG"), mask("GGT")))  # the paths of A(C+|G+)T at k = 3
        kms = ("ACT", "AGT", "ACC", "GGT", "CCT", "AGG") if quirk else ("ACC",)
        out[name] = {"n_max": n_max, "bin_size": m, "kmer_masks": {k: bits(k) for k in kms}, "candidate_bins": [b for b in range(5) if cand[b]]}
    fx = lambda kmer: [all((fixture_words[hash_row(canonical(pack_dna(kmer), 3), SEEDS[i], 64)] >> b) & 1 for i in range(3)) for b in range(2)]
    s = lambda m: "".join("1" if x else "0" for x in reversed(m))
    acg, acc, ccg = fx("ACG"), fx("ACC"), fx("CCG")
    cand = [a or (b and c) for a, b, c in zip(acg, acc, ccg)]  # AC+G: ACG | ACC.CCG
    out["fixture_query"] = {"regex": "AC+G", "masks": {"ACG": s(acg), "ACC": s(acc), "CCG": s(ccg)}, "candidate": s(cand)}
    return out


def config2():
    base = dict(zip("ABCDEFGHIJKLMNOPQRSTUVWXYZ", [0, 2, 1, 2, 3, 4, 5, 6, 7, 9, 8, 9, 10, 11, 20, 12, 13, 14, 15, 16, 20, 17, 18, 20, 19, 3]))
    pack = lambda s: sum(base[c] << (5 * (len(s) - 1 - i)) for i, c in enumerate(s))
    kmers = ["LMAE", "MAEG", "AEGL", "EGLY", "GLYN", "LMAQ", "MAQG", "AQGL", "QGLY"]
    return {"motif": "LMA(E|Q)GLYN", "k": 4, "kmers": {k: pack(k) for k in kmers}}


def main():
    check = "--check" in sys.argv
    kat, words = hash_kat()
    files = {"hash_kat.json": kat, "config1_masks.json": config1(words), "config2_kmers.json": config2()}
    bad = 0
    for name, data in files.items():
        path = os.path.join(HERE, name)
        if check:
            if json.load(open(path)) != data:
                print("DIFFERS:", name)
                bad += 1
        else:
            with open(path, "w") as f:
                json.dump(data, f, indent=1)
                f.write("\n")
    if check:
        print("golden files regenerate identically" if not bad else "%d golden file(s) differ" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
