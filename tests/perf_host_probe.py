#!/usr/bin/env python3
"""Secondary measurement: the PCIe-inclusive rate of the host-buffer entry point txq_probe
(k-mers and masks in pageable host memory), for DESIGN.md.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    from helpers import random_words, splitmix64
    from tetrex_amd import capi
    capi.init(0)
    bins, m, h = 1024, 1247045, 3
    ix = capi.Index.create_ibf(bins, m, h)
    n = 1 << 22
    vals = splitmix64(1, n) >> np.uint64(44)
    dv = capi.DeviceBuffer.from_numpy(vals)
    db = capi.DeviceBuffer.from_numpy((splitmix64(2, n) % np.uint64(bins)).astype(np.uint32))
    ix.emplace_device(dv.ptr, db.ptr, n)
    capi.synchronize()
    kmers = splitmix64(3, n) >> np.uint64(44)
    ix.probe(kmers[:1 << 16])
    res = {"entry_point": "txq_probe (host buffers)", "kmers": n, "bytes_moved_per_probe": 8 + 128}
    pageable = np.ones((n, 16), dtype=np.uint64)   # pages already touched
    pinned = capi.HostBuffer((n, 16))
    for name, out in (("pageable", pageable), ("pinned", pinned.array)):
        ix.probe(kmers, out=out)
        t0 = time.perf_counter()
        for _ in range(3):
            ix.probe(kmers, out=out)
        dt = (time.perf_counter() - t0) / 3
        res[name] = {"seconds": dt, "probes_per_s": n / dt, "effective_GBps": n * 136 / dt / 1e9}
    assert np.array_equal(pageable, pinned.array)
    res["nonzero_masks"] = int(pageable.any(axis=1).sum())
    pinned.free()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
