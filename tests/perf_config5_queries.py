#!/usr/bin/env python3
"""Secondary measurement: whole queries on a BASELINE configs[4]-shaped index — 65536-user-bin HIBF
(256 x 256, h = 2), Murphy-reduced peptide alphabet, k = 5 — where every mask is 8 KiB.  The tree is
built on the device like bench.py's HIBF leg (no oracle involved); prints one JSON line.  Used with
rocprofv3 to see how the mask-DAG executor behaves at 1024-word masks."""
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
    user_bins, children, per_bin, h = 65536, 256, 300, 2
    per_child = user_bins // children
    rng = np.random.default_rng(5)
    shifts = np.uint64(5) * np.arange(4, -1, -1, dtype=np.uint64)

    def values(count):
        return (rng.integers(0, 10, size=(count, 5)).astype(np.uint64) << shifts).sum(axis=1).astype(np.uint64)

    def filled(bins, rows, vals, bins_of):
        ix = capi.Index.create_ibf(bins, rows, h)
        dv = torch.from_numpy(vals.view(np.int64)).cuda()
        db = torch.from_numpy(bins_of.astype(np.uint32).view(np.int32)).cuda()
        ix.emplace_device(dv.data_ptr(), db.data_ptr(), vals.size, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        words = ix.download_words_rows(rows)
        ix.free()
        return words

    m_child = bench.compute_bitcount(per_bin, 0.05)
    m_root = bench.compute_bitcount(per_bin * per_child, 0.05)
    tb_of = np.repeat(np.arange(per_child, dtype=np.uint32), per_bin)
    descs, rv, rb = [None], [], []
    for c in range(children):
        v = values(per_child * per_bin)
        descs.append(dict(bins=per_child, bin_size=m_child, hash_funs=h, words=filled(per_child, m_child, v, tb_of),
                          next_ibf_id=np.zeros(per_child, dtype=np.uint64), tb_to_user=np.arange(c * per_child, (c + 1) * per_child, dtype=np.uint64)))
        rv.append(v)
        rb.append(np.full(v.size, c, dtype=np.uint32))
    descs[0] = dict(bins=children, bin_size=m_root, hash_funs=h, words=filled(children, m_root, np.concatenate(rv), np.concatenate(rb)),
                    next_ibf_id=np.arange(1, children + 1, dtype=np.uint64), tb_to_user=np.full(children, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64))
    ix = capi.Index.upload_hibf(user_bins, descs)
    out = {"workload": "queries on S-HIBF-65536, Murphy alphabet, k=5, 8 KiB masks"}
    for name, motifs in (("plain", random_prosite_motifs(200, 7, wildcard=0.0, classes=0.3, ranges=0.0)),
                         ("wildcards", random_prosite_motifs(200, 6))):
        ix.query_masks(motifs[:5], False, 5, 1)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            masks, status, stats = ix.query_masks(motifs, False, 5, 1)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, stats)
        out[name] = {"motifs": len(motifs), "seconds": best[0], "queries_per_s": len(motifs) / best[0],
                     "failed": int(sum(1 for s in status if s)), **best[1]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
