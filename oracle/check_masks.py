#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (oracle/): compare candidate-bin masks with the CPU oracle, many queries at once.

bench.py's parity legs start this as a CHILD process (no GPU in it) so that the comparison can use all host threads and be
stopped at a deadline even in the middle of a query the oracle needs minutes for (it enumerates every state: a motif that
begins with wildcards at k = 6).  Input: a directory with
  meta.json   {"kind": "ibf" | "hibf", "bins", "rows", "h", "dna", "k", "reduction", "threads", "motifs": [...]}
  index.npz   ibf: words;  hibf: n, and per IBF i  bins_i, rows_i, h_i, words_i, next_i, user_i
  masks.npy   [len(motifs), words] the masks to check (uint64)
Output, one line per query as it finishes (in that order):  "ok <i>" | "MISMATCH <i>" | "refused <i>" (the reference path cannot search it).
Masks are compared with Index.expected_mask: the reference's restated result, or — where that is implementation-defined
(quirk merges) — the result under well-defined merges."""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor, as_completed

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import oracle as O
    d = sys.argv[1]
    meta = json.load(open(os.path.join(d, "meta.json")))
    z = np.load(os.path.join(d, "index.npz"))
    masks = np.load(os.path.join(d, "masks.npy"))
    if meta["kind"] == "ibf":
        ox = O.Index.ibf(meta["bins"], meta["rows"], meta["h"], dna=meta["dna"], k=meta["k"], reduction=meta.get("reduction", 0))
        ox.set_words(z["words"])
    else:
        ox = O.Index.hibf(meta["bins"], dna=meta["dna"], k=meta["k"], reduction=meta.get("reduction", 0))
        for i in range(int(z["n"])):
            ox.add_ibf(int(z["bins_%d" % i]), int(z["rows_%d" % i]), int(z["h_%d" % i]), z["next_%d" % i], z["user_%d" % i], words=z["words_%d" % i])
    motifs = meta["motifs"]

    def one(i):
        try:
            want = ox.expected_mask(motifs[i])[0]
        except Exception:  # noqa: BLE001 - a motif the reference path cannot search either
            return "refused %d" % i
        return ("ok %d" if np.array_equal(want, masks[i]) else "MISMATCH %d") % i

    # (lines in the order the queries FINISH: one the oracle needs minutes for must not hold back the verdicts of those after it
    # when the parent stops this process at its deadline)
    with ThreadPoolExecutor(max_workers=int(meta.get("threads", 1))) as pool:
        for done in as_completed([pool.submit(one, i) for i in range(len(motifs))]):
            print(done.result(), flush=True)


if __name__ == "__main__":
    main()
