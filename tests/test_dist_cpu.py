"""The N>1 path on CPU: world_size-2/3 gloo process groups exercising the shard arithmetic and
the final-mask all-gather (tetrex_amd/dist.py).  The per-shard device output is checked against
the same column slices in tests/test_gpu_probe.py / test_gpu_query.py; here the oracle's full
masks are sliced exactly like the device shards would be and must reassemble bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bins, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as O
        from helpers import random_words, splitmix64
        from tetrex_amd.dist import shard_range, gather_final_masks, or_reduce_alive
        m, h = 509, 3
        words = random_words(bins, m, 0.12, 7)
        ox = O.Index.ibf(bins, m, h, dna=False, k=4)
        ox.set_words(words)
        kmers = splitmix64(3, n) >> np.uint64(44)
        full = ox.probe(kmers)                       # what an unsharded index answers
        W = full.shape[1]
        lo, hi = shard_range(W, rank, world)
        local = torch.from_numpy(full[:, lo:hi].copy().view(np.int64))
        got = gather_final_masks(local, W).numpy().view(np.uint64)
        ok = np.array_equal(got, full)
        # alive: OR over shards of "my columns are non-zero" == "the full mask is non-zero"
        alive_local = np.packbits(full[:, lo:hi].any(axis=1) if hi > lo else np.zeros(n, bool), bitorder="little")
        alive = or_reduce_alive(torch.from_numpy(alive_local)).numpy()
        ok = ok and np.array_equal(alive, np.packbits(full.any(axis=1), bitorder="little"))
        q.put((rank, bool(ok), int(lo), int(hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,bins", [(2, 1024), (2, 300), (3, 1024), (2, 40)])
def test_sharded_masks_reassemble_over_gloo(world, bins):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bins, 333, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res), res
    W = (bins + 63) // 64
    assert res[0][2] == 0 and res[-1][3] == W
    assert all(res[i][3] == res[i + 1][2] for i in range(world - 1))  # shards tile the mask


def test_shard_range_matches_the_device_library():
    from tetrex_amd.dist import shard_range
    for words in (0, 1, 2, 5, 16, 47, 128, 141):
        for world in (1, 2, 3, 8, 17):
            cover = []
            for r in range(world):
                lo, hi = shard_range(words, r, world)
                assert 0 <= lo <= hi <= words and hi - lo in (words // world, words // world + 1)
                cover += list(range(lo, hi))
            assert cover == list(range(words))
