#!/usr/bin/env python3
"""Secondary measurement: BASELINE configs[2] — a 1024-user-bin peptide HIBF (16 children of 64 bins, k = 4, h = 3) and
the bench's batch of 1000 PROSITE-style motifs (10 % wildcards, 5 % x(m,n)) streamed to one MI355X, next to the same
batch on the flat 1024-bin IBF built from the same values.  Built on the device (no oracle involved); the HIBF masks
are checked against the flat index's: a user bin the flat IBF rules out... may still pass the HIBF (different filters),
so only the property both share is checked: every bin that holds all k-mers of a literal motif is a candidate in both.
Prints one JSON line.  TETREX_DENSE=0 gives the round-1 behaviour (enumerated states) for the A/B."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import torch
    import bench
    from motifs import random_prosite_motifs
    from tetrex_amd import capi
    capi.init(0)
    user_bins, children, per_bin, h = 1024, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 3
    per_child = user_bins // children
    rng = np.random.default_rng(5)

    def filled(bins, rows, vals, bins_of):
        ix = capi.Index.create_ibf(bins, rows, h)
        dv = torch.from_numpy(vals.view(np.int64)).cuda()
        db = torch.from_numpy(bins_of.astype(np.uint32).view(np.int32)).cuda()
        ix.emplace_device(dv.data_ptr(), db.data_ptr(), vals.size, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        words = ix.download_words_rows(rows)
        ix.free()
        return words

    values = [rng.integers(0, 1 << 20, size=per_bin, dtype=np.uint64) for _ in range(user_bins)]
    m_child = bench.compute_bitcount(per_bin, 0.05)
    m_root = bench.compute_bitcount(per_bin * per_child, 0.05)
    descs, rv, rb = [None], [], []
    for c in range(children):
        v = np.concatenate(values[c * per_child:(c + 1) * per_child])
        descs.append(dict(bins=per_child, bin_size=m_child, hash_funs=h, words=filled(per_child, m_child, v, np.repeat(np.arange(per_child, dtype=np.uint32), per_bin)),
                          next_ibf_id=np.zeros(per_child, dtype=np.uint64), tb_to_user=np.arange(c * per_child, (c + 1) * per_child, dtype=np.uint64)))
        rv.append(v)
        rb.append(np.full(v.size, c, dtype=np.uint32))
    descs[0] = dict(bins=children, bin_size=m_root, hash_funs=h, words=filled(children, m_root, np.concatenate(rv), np.concatenate(rb)),
                    next_ibf_id=np.arange(1, children + 1, dtype=np.uint64), tb_to_user=np.full(children, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64))
    hibf = capi.Index.upload_hibf(user_bins, descs)
    flat = capi.Index.upload_ibf(user_bins, m_child, h, filled(user_bins, m_child, np.concatenate(values), np.repeat(np.arange(user_bins, dtype=np.uint32), per_bin)))
    out = {"workload": "BASELINE configs[2]: 1000 PROSITE-style motifs on a 1024-user-bin HIBF (16 x 64, k=4, h=3), %d values per bin" % per_bin,
           "dense_steps": os.environ.get("TETREX_DENSE", "1") != "0"}
    motifs = random_prosite_motifs(1000, 6)
    for name, ix in (("hibf", hibf), ("flat_ibf_same_values", flat)):
        ix.query_masks(motifs[:10], False, 4)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            masks, status, stats = ix.query_masks(motifs, False, 4)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, stats, masks)
        out[name] = {"motifs": len(motifs), "seconds": best[0], "queries_per_s": len(motifs) / best[0], "failed": int(sum(1 for s in status if s)),
                     "mean_candidate_bins": float(np.unpackbits(best[2].view(np.uint8), axis=1).sum(axis=1).mean()), **best[1]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
