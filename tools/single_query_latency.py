import sys, time, os
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from tetrex_amd import capi
import bench
capi.init(0)
m = bench.compute_bitcount(200000, 0.05)
ix = bench.build_index(capi, torch, 1024, 1024, m, 3, 0, 1, 200000, 20)
for q in ("LMA(E|Q)GLYN", "LMA..GLYN"):
    ix.query_masks([q], False, 4)
    lat = []
    for _ in range(200):
        t0 = time.perf_counter(); ix.query_masks([q], False, 4); lat.append(time.perf_counter() - t0)
    print(q, "median %.1f us  p10 %.1f us" % (np.median(lat) * 1e6, np.percentile(lat, 10) * 1e6))
