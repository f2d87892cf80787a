"""Which k-mers does the layout-order path of the hibf-layout-900 tree get wrong?  Every 4-mer as a literal query (no table of
k-mer masks) against txq_probe (user order, descent kernels)."""
import os, sys, warnings, itertools
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
warnings.filterwarnings("ignore")
os.environ["TXQ_KMER_TABLE_MB"] = "0"
import oracle as O
from tetrex_amd import capi
from helpers import layout_hibf
capi.init(0)
ox, descs, values = layout_hibf(O, 5, user_bins=900, tmax=32, n_values=30)
ix = capi.Index.upload_hibf(900, descs)
A = "ACDEFGHIKLMNPQRSTVWY"
code = {c: i for i, c in enumerate(A)}
bad = 0
all4 = ["".join(t) for t in itertools.product(A, repeat=4)]
vals = np.array([sum(code[c] << (5 * (3 - j)) for j, c in enumerate(s)) for s in all4], dtype=np.uint64)
user = ix.probe(vals)
for at in range(0, len(all4), 20000):
    part = all4[at:at + 20000]
    got, status, stats = ix.query_masks(part, False, 4)
    diff = np.nonzero((got != user[at:at + 20000]).any(axis=1))[0]
    for d in diff[:10]:
        g, w = got[d], user[at + d]
        x = np.unpackbits((g ^ w).view(np.uint8), bitorder="little")
        print("k-mer %s: layout path %d bits, user order %d bits, differ in bins %s" % (part[d], int(np.unpackbits(g.view(np.uint8)).sum()), int(np.unpackbits(w.view(np.uint8)).sum()), np.nonzero(x)[0][:8].tolist()))
    bad += len(diff)
print("k-mers that differ:", bad, "of", len(all4))
# the oracle on a sample of the differing ones is the referee
