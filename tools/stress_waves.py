#!/usr/bin/env python3
"""Stress of the pipelined session (waves of queries, two staging sets, two streams — stages of consecutive waves running
BESIDE each other —, blocks given back and reused, the index's table of all k-mers' masks): a batch under random wave sizes,
task budgets, block pools, dense thresholds and table settings, every run's masks against those of the plainest run (one wave,
one stream, rows gathered).  k = 4: saturated lists as full blocks (dense_kernel); k = 6: tracked blocks with live lists
(sparse_kernel), on a smaller index than the bench's.
Usage on the GPU box: tools/stress_waves.py [iterations] [k]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from tetrex_amd import capi
import bench
from motifs import random_prosite_motifs

capi.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
if k == 4:
    m = bench.compute_bitcount(200000, 0.05)
    ix = bench.build_index(capi, torch, 1024, 1024, m, 3, 0, 1, 200000, 20)
    motifs = random_prosite_motifs(600, 123)
    os.environ["TETREX_DENSE_EVIDENCE"] = "dense"
else:
    bins, per_bin, h = 1024, 50000, 3
    m = bench.compute_bitcount(per_bin, 0.05)
    ix = capi.Index.create_ibf(bins, m, h)
    rng0 = np.random.default_rng(11)
    for b0 in range(0, bins, 128):
        codes = rng0.integers(0, 20, size=(128, per_bin)).astype(np.uint64)
        vals = np.zeros((128, per_bin - k + 1), dtype=np.uint64)
        for j in range(k):
            vals = (vals << np.uint64(5)) | codes[:, j:per_bin - k + 1 + j]
        bins_of = np.repeat(np.arange(b0, b0 + 128, dtype=np.uint32), vals.shape[1])
        dv = torch.from_numpy(vals.reshape(-1).view(np.int64)).cuda()
        db = torch.from_numpy(bins_of.view(np.int32)).cuda()
        ix.emplace_device(dv.data_ptr(), db.data_ptr(), vals.size, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    motifs = random_prosite_motifs(300, 3, wildcard=0.05, ranges=0.02, min_len=8, max_len=14)
    os.environ["TETREX_DENSE_EVIDENCE"] = "thin"
os.environ.update(TETREX_WAVE_OPS="0", TXQ_ONE_STREAM="1", TXQ_KMER_TABLE_MB="0")
want, status, plain_stats = ix.query_masks(motifs, False, k)
del os.environ["TXQ_ONE_STREAM"]
rng = np.random.default_rng(7)
most = max(2000, plain_stats["ops"] // 2)
for it in range(n):
    os.environ["TETREX_WAVE_OPS"] = str(int(rng.integers(300, most)))
    if rng.random() < 0.4:
        os.environ["TETREX_TASK_OPS"] = str(int(rng.integers(100, 5000)))
    else:
        os.environ.pop("TETREX_TASK_OPS", None)
    os.environ["TETREX_DENSE_POOL_MB"] = str(int(rng.choice([200, 2000, 49152])))
    os.environ["TXQ_KMER_TABLE_MB"] = str(int(rng.choice([0, 512])))
    if rng.random() < 0.3:
        os.environ["TETREX_DENSE_MIN"], os.environ["TETREX_DENSE_SPARSE_BELOW"] = "32", "16"
    else:
        os.environ.pop("TETREX_DENSE_MIN", None); os.environ.pop("TETREX_DENSE_SPARSE_BELOW", None)
    got, st, stats = ix.query_masks(motifs, False, k)
    assert list(st) == list(status), it
    assert np.array_equal(got, want), (it, {a: b for a, b in os.environ.items() if a.startswith(("TETREX_", "TXQ_"))})
    print("it %d: wave %s task %s pool %s table %s min %s -> %d stages, %d ops, %d dense ops, %d tracked: masks equal" % (
        it, os.environ["TETREX_WAVE_OPS"], os.environ.get("TETREX_TASK_OPS"), os.environ["TETREX_DENSE_POOL_MB"], os.environ["TXQ_KMER_TABLE_MB"],
        os.environ.get("TETREX_DENSE_MIN"), stats["stages"], stats["ops"], stats["dense_ops"], stats["tracked_queries"]), flush=True)
print("stress ok: %d runs at k = %d" % (n, k))
