#!/usr/bin/env python3
"""Secondary measurement (not the contract bench; lives under tests/ because it builds its tree with
the oracle and checks a sample against it): HIBF descent rate on a BASELINE configs[4]-shaped
tree (65536 user bins, 256-wide root of merged bins over 256 children, h=2, Murphy k=5 values).
Prints one JSON line.  Used with rocprofv3 for the per-kernel table in DESIGN.md."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import oracle as O
    from helpers import regular_hibf
    from tetrex_amd import capi
    n_probe = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
    per_bin = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    n_shards = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # >1: time shard 0 of a column-sharded upload (one rank's work)
    user_bins = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
    children = int(sys.argv[5]) if len(sys.argv) > 5 else 256
    rng = np.random.default_rng(5)
    shifts = np.uint64(5) * np.arange(4, -1, -1, dtype=np.uint64)

    def vals(b):
        return (rng.integers(0, 10, size=(per_bin, 5)).astype(np.uint64) << shifts).sum(axis=1).astype(np.uint64)
    t0 = time.perf_counter()
    ox, descs, values = regular_hibf(O, user_bins, children, per_bin, vals, h=2, k=5, reduction=1)
    t_build = time.perf_counter() - t0
    capi.init(0)
    ix = capi.Index.upload_hibf(user_bins, descs, shard_rank=0, n_shards=n_shards)
    W = ix.shard_words
    # half of the probes are inserted values (they descend to a leaf), half are random (mostly stop at the root)
    present = np.concatenate([v[:8] for v in values[::8]])
    kmers = np.concatenate([np.resize(present, n_probe // 2), (rng.integers(0, 10, size=(n_probe // 2, 5)).astype(np.uint64) << shifts).sum(axis=1).astype(np.uint64)])
    rng.shuffle(kmers)
    dk = capi.DeviceBuffer.from_numpy(kmers)
    dm = capi.DeviceBuffer(n_probe * W * 8)
    for _ in range(2):
        ix.probe_device(dk.ptr, n_probe, dm.ptr)
    capi.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        ix.probe_device(dk.ptr, n_probe, dm.ptr)
    capi.synchronize()
    dt = (time.perf_counter() - t0) / reps
    sample = 2000
    got = dm.to_numpy(np.uint64, (n_probe, W))[:sample]
    if not os.environ.get("PERF_HIBF_NO_CHECK"):  # timing experiments with deliberately wrong kernels set this
        assert np.array_equal(got, ox.probe(kmers[:sample])[:, :W]), "HIBF masks differ from the oracle"
    print(json.dumps({"workload": "S-HIBF-%d" % user_bins, "n_shards": n_shards, "user_bins": user_bins, "n_ibf": int(ix.info.n_ibf), "kmers": n_probe,
                      "seconds_per_batch": dt, "kmers_per_s": n_probe / dt, "mask_bytes_per_kmer": W * 8,
                      "mask_zero_fill_GBps": n_probe * W * 8 / dt / 1e9, "device_bytes": int(ix.info.device_bytes),
                      "index_build_s": round(t_build, 1), "parity_sample": sample}))


if __name__ == "__main__":
    main()
