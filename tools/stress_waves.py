#!/usr/bin/env python3
"""Stress of the pipelined session (waves of queries, two staging sets, two streams, recycled regions): the bench batch
under random wave sizes and task budgets, every run's masks against those of the plainest run (one wave, one stream).
Usage on the GPU box: tools/stress_waves.py [iterations]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from tetrex_amd import capi
import bench
from motifs import random_prosite_motifs

capi.init(0)
m = bench.compute_bitcount(200000, 0.05)
ix = bench.build_index(capi, torch, 1024, 1024, m, 3, 0, 1, 200000, 20)
motifs = random_prosite_motifs(600, 123)
os.environ.update(TETREX_WAVE_OPS="0", TXQ_ONE_STREAM="1", TETREX_DENSE_EVIDENCE="dense")
want, status, _ = ix.query_masks(motifs, False, 4)
del os.environ["TXQ_ONE_STREAM"]
rng = np.random.default_rng(7)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
beside = 0
for it in range(n):
    os.environ["TETREX_WAVE_OPS"] = str(int(rng.integers(300, 60000)))
    if rng.random() < 0.4:
        os.environ["TETREX_TASK_OPS"] = str(int(rng.integers(100, 5000)))
    else:
        os.environ.pop("TETREX_TASK_OPS", None)
    os.environ["TETREX_DENSE_POOL_MB"] = str(int(rng.choice([200, 2000, 49152])))
    got, st, stats = ix.query_masks(motifs, False, 4)
    assert list(st) == list(status), it
    assert np.array_equal(got, want), (it, dict(os.environ))
    print("it %d: wave %s task %s pool %s -> %d stages, %d dense ops: masks equal" % (it, os.environ["TETREX_WAVE_OPS"], os.environ.get("TETREX_TASK_OPS"),
                                                                                   os.environ["TETREX_DENSE_POOL_MB"], stats["stages"], stats["dense_ops"]), flush=True)
print("stress ok: %d runs" % n)
