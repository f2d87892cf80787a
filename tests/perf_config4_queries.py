#!/usr/bin/env python3
"""Secondary measurement: a 10k-motif DNA batch on ONE column shard of a BASELINE configs[3]-shaped index
(8192-bin DNA IBF of 62.5 M rows sharded 8 ways: this GPU holds 1024 bins x 62.5 M rows = 8 GB), k = 16.
Every bin gets the canonical 16-mers of a random 100 kb sequence; the motifs are 24-32 nt windows of those
sequences with a few positions turned into wildcards, classes and small gaps, so every motif has a bin it
must be found in (checked).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    from tetrex_amd import capi, host
    n_motifs = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    bins, rows, h, k, seq_len = 1024, 62500000, 3, 16, 100000
    capi.init(0)
    rng = np.random.default_rng(4)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    ix = capi.Index.create_ibf(bins, rows, h)
    seqs = []
    t0 = time.perf_counter()
    for b0 in range(0, bins, 64):
        vals, bins_of = [], []
        for b in range(b0, min(bins, b0 + 64)):
            s = acgt[rng.integers(0, 4, size=seq_len)].tobytes()
            seqs.append(s)
            v = host.record_values_array(s, k, dna=True)
            vals.append(v)
            bins_of.append(np.full(v.size, b, dtype=np.uint32))
        v = np.concatenate(vals)
        dv, db = capi.DeviceBuffer.from_numpy(v), capi.DeviceBuffer.from_numpy(np.concatenate(bins_of))
        ix.emplace_device(dv.ptr, db.ptr, v.size)
        capi.synchronize()
    build_s = time.perf_counter() - t0
    motifs, home = [], []
    for i in range(n_motifs):
        b = int(rng.integers(0, bins))
        L = int(rng.integers(24, 33))
        at = int(rng.integers(0, seq_len - L))
        w = list(seqs[b][at:at + L].decode())
        for p in rng.choice(np.arange(4, L - 4), size=int(rng.integers(0, 3)), replace=False):  # up to two degenerate positions away from the ends
            p = int(p)
            r = rng.random()
            if r < 0.4:
                w[p] = "."
            elif r < 0.8:
                w[p] = "[" + "".join(sorted(set(w[p] + "ACGT"[int(rng.integers(0, 4))]))) + "]"
            else:
                w[p] = w[p] + "?"  # the residue may be missing
        motifs.append("".join(w))
        home.append(b)
    ix.query_masks(motifs[:50], True, k)
    t0 = time.perf_counter()
    masks, status, stats = ix.query_masks(motifs, True, k)
    dt = time.perf_counter() - t0
    ok = [s == 0 for s in status]
    found = [(int(masks[i, home[i] >> 6]) >> (home[i] & 63)) & 1 for i in range(n_motifs) if ok[i]]
    assert all(found), "a motif lost the bin it was taken from"
    print(json.dumps({"workload": "DNA k=16 motif batch on one 1024-bin x 62.5 M-row shard (8 GB) of the 8192-bin index", "motifs": n_motifs,
                      "seconds": dt, "queries_per_s": n_motifs / dt, "failed": int(n_motifs - sum(ok)), **stats,
                      "mean_candidate_bins": float(np.unpackbits(masks.view(np.uint8), axis=1).sum(axis=1).mean()),
                      "index_build_s": round(build_s, 1), "device_bytes": int(ix.info.device_bytes)}))


if __name__ == "__main__":
    main()
