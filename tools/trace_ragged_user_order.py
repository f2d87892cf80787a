import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import oracle as O
from helpers import layout_hibf
from motifs import random_prosite_motifs
from tetrex_amd import capi
capi.init(0)
ox, descs, values = layout_hibf(O, 9, user_bins=65536, tmax=256, n_values=12)
ix = capi.Index.upload_hibf(65536, descs)
motifs = random_prosite_motifs(200, 3, wildcard=0.08, ranges=0.04, min_len=6, max_len=12)
os.environ["TXQ_HIBF_LAYOUT_ORDER"] = "0"
ix.query_masks(motifs[:20], False, 4)
os.environ["TXQ_TRACE"] = "1"; os.environ["TETREX_TRACE"] = "1"
t = time.perf_counter(); m, st, stats = ix.query_masks(motifs, False, 4); print("user order", time.perf_counter() - t, stats, flush=True)
