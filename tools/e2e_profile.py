import sys, time, os, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from tetrex_amd import capi
import bench
from motifs import random_prosite_motifs
capi.init(0)
class A: pass
m = bench.compute_bitcount(200000, 0.05)
ix = bench.build_index(capi, torch, 1024, 1024, m, 3, 0, 1, 200000, 20)
motifs = random_prosite_motifs(int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 6)
ix.query_masks(motifs[:10], False, 4)
for rep in range(int(os.environ.get("REPS", "6"))):
    t0 = time.perf_counter()
    masks, status, stats = ix.query_masks(motifs, False, 4)
    dt = time.perf_counter() - t0
    print("rep", rep, "%.2f ms" % (dt * 1e3), stats, flush=True)
