#!/usr/bin/env python3
"""Measurement: a GENERAL HIBF of 65 536 user bins (helpers.layout_hibf: at most `tmax` technical bins per IBF, user bins
scattered over the leaves, split bins, user bins next to merged bins — three levels at tmax = 64), the shape seqan::hibf's
layout gives the reference's index (include/index_hibf.h:114-129).
  * txq_probe_device (masks in user-bin order, the public contract): the k-mer-stationary descent kernels;
  * the same k-mers as ONE session stage (rows in layout order: csrc/txq_hibf.hip hibf_layout_level_kernel) — what every
    query on such an index runs on;
  * a batch of PROSITE-style motifs, in layout order and (TXQ_HIBF_LAYOUT_ORDER=0) in user-bin order; masks must agree.
Prints one JSON line.  Usage: perf_hibf_ragged.py [kmers] [tmax] [user_bins]"""
import json
import os
import struct
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def probe_blob(kmers):
    """a version-2 blob (include/txq_program.h): the k-mer table and one program with the op RESULT |= ONES & M[0]"""
    n = kmers.size
    k_off = 64
    p_off = k_off + n * 8
    o_off = p_off + 24
    l_off = o_off + 16
    head = struct.pack("<6I5Q", 0x50515854, 2, 1, n, 1, 1, k_off, p_off, o_off, l_off, 0)
    return head + kmers.tobytes() + struct.pack("<6I", 0, 1, 3, 0, 1, 0) + struct.pack("<4I", 0, 2, 1, 2) + struct.pack("<2I", 1, 0)


def measure_sharded(capi, torch, dist, coll_device, rank, world, n=1 << 18, tmax=256, user_bins=65536):
    """N > 1 (bench.py under torch.distributed.run): the general tree sharded by SUB-TREES (txq_index_upload_subtrees: the root
    replicated, its sub-trees dealt over the ranks) — every rank works in layout order on its own part of the tree and emits
    full-width masks that are ORed (tetrex_amd/dist.py or_join_final_masks; split bins may straddle ranks).  The 200-motif batch
    per rank, the join, and on rank 0 a sample of the joined masks against the CPU oracle."""
    import oracle as O
    from helpers import layout_hibf
    from motifs import random_prosite_motifs
    from tetrex_amd.dist import or_join_final_masks
    motifs = random_prosite_motifs(200, 3, wildcard=0.08, ranges=0.04, min_len=6, max_len=12)
    err = ix = None
    try:  # rank-local work first: a failure here must not leave the other ranks waiting in a collective
        ox, descs, values = layout_hibf(O, 9, user_bins=user_bins, tmax=tmax, n_values=12)
        ix = capi.Index.upload_hibf(user_bins, descs, shard_rank=rank, n_shards=world, subtrees=True)
        assert int(ix.info.join_or) == 1
        ix.query_masks(motifs[:20], False, 4)
    except Exception as e:  # noqa: BLE001
        err = repr(e)
    dist.barrier()
    t = time.perf_counter()
    masks = status = stats = None
    if err is None:
        try:
            masks, status, stats = ix.query_masks(motifs, False, 4)
        except Exception as e:  # noqa: BLE001
            err = repr(e)
    local_s = time.perf_counter() - t
    ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=coll_device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        if ix is not None:
            ix.free()
        return {"error": err or "another rank failed"}
    tj = time.perf_counter()
    full = or_join_final_masks(torch.from_numpy(masks.view(np.int64)).to(coll_device))
    torch.cuda.synchronize()
    join_s = time.perf_counter() - tj
    tt = torch.tensor([local_s + join_s], dtype=torch.float64, device=coll_device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    total = float(tt.item())
    full = full.cpu().numpy().view(np.uint64)
    out = {"tree": "general HIBF, %d user bins, tmax %d: %d IBFs, sharded by sub-trees over %d ranks" % (user_bins, tmax, len(descs), world),
           "ibfs_of_this_rank": int(ix.info.n_ibf), "queries_layout_order": {"seconds": total, "queries_per_s": len(motifs) / total, "join_seconds": join_s, **stats},
           "collective": {"backend": dist.get_backend(), "ranks": world, "op": "all_gather of the full-width masks (%d x %d words per rank), ORed" % (len(motifs), masks.shape[1])}}
    if rank == 0:
        compared, t0 = 0, time.perf_counter()
        for q, g, st in zip(motifs, full, status):
            if st:
                continue
            want, ost = ox.query(q, with_stats=True)
            if not ost["quirk_merges"]:
                if not np.array_equal(g, want):
                    raise SystemExit("sub-tree shards: the joined mask of %r differs from the oracle" % q)
                compared += 1
            if time.perf_counter() - t0 > 5.0:
                break
        out["oracle_masks_compared"] = compared
    ix.free()
    return out


def measure(capi, torch, n=1 << 20, tmax=64, user_bins=65536):
    """the measurement (bench.py's `hibf_irregular` leg calls it with its own capi / torch); returns the dict that main() prints"""
    import oracle as O
    from helpers import layout_hibf
    from motifs import random_prosite_motifs
    t0 = time.perf_counter()
    ox, descs, values = layout_hibf(O, 9, user_bins=user_bins, tmax=tmax, n_values=12)
    build_s = time.perf_counter() - t0
    ix = capi.Index.upload_hibf(user_bins, descs)
    rng = np.random.default_rng(3)
    kmers = np.concatenate([np.concatenate([v[:2] for v in values[:4096]]), rng.integers(0, 1 << 20, size=n - 8192, dtype=np.uint64)]) if n > 8192 else rng.integers(0, 1 << 20, size=n, dtype=np.uint64)
    W = ix.shard_words
    d_kmers = torch.from_numpy(kmers.view(np.int64)).cuda()
    d_masks = torch.empty((n, W), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def user_order():
        ix.probe_device(d_kmers.data_ptr(), n, d_masks.data_ptr(), 0, stream)
        torch.cuda.synchronize()
    queries_only = bool(os.environ.get("PERF_HIBF_QUERIES_ONLY"))  # (profiling runs: the layout-order motif batch only)
    if queries_only:
        del d_masks
        motifs = random_prosite_motifs(200, 3, wildcard=0.08, ranges=0.04, min_len=6, max_len=12)
        ix.query_masks(motifs[:20], False, 4)
        for _ in range(4):
            t = time.perf_counter()
            ix.query_masks(motifs, False, 4)
            dt = time.perf_counter() - t
        ix.free()
        return {"queries_layout_order": {"seconds": dt}}
    user_order()
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        user_order()
        ts.append(time.perf_counter() - t)
    t_user = min(ts)
    got = d_masks[:2048].cpu().numpy().view(np.uint64)
    if not os.environ.get("PERF_HIBF_NO_CHECK") and not np.array_equal(got, ox.probe(kmers[:2048])):
        raise SystemExit("user-order masks differ from the oracle")
    del d_masks
    blob = probe_blob(kmers)
    ix.run_programs(blob, 1)
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        res = ix.run_programs(blob, 1)
        ts.append(time.perf_counter() - t)
    t_layout = min(ts)
    want0 = ox.probe(kmers[:1])[0]
    if not os.environ.get("PERF_HIBF_NO_CHECK") and not np.array_equal(res[0], want0):
        raise SystemExit("layout-order stage: RESULT differs from the oracle's mask of the first k-mer")
    info = ix.info
    if os.environ.get("PERF_HIBF_PROBE_ONLY"):  # (profiling runs: the probes only)
        ix.free()
        return {"probe_user_order": {"seconds": t_user, "kmers_per_s": n / t_user}, "probe_layout_order": {"seconds": t_layout, "kmers_per_s": n / t_layout}}
    motifs = random_prosite_motifs(200, 3, wildcard=0.08, ranges=0.04, min_len=6, max_len=12)
    timings = {}
    masks = {}
    for way in ("layout", "user"):
        if way == "user":
            os.environ["TXQ_HIBF_LAYOUT_ORDER"] = "0"
        else:
            os.environ.pop("TXQ_HIBF_LAYOUT_ORDER", None)
        ix.query_masks(motifs[:20], False, 4)
        best = None
        for _ in range(3):
            t = time.perf_counter()
            m, status, stats = ix.query_masks(motifs, False, 4)
            dt = time.perf_counter() - t
            if best is None or dt < best[0]:
                best = (dt, stats)
        timings[way] = {"seconds": best[0], "queries_per_s": len(motifs) / best[0], **best[1]}
        masks[way] = (m, list(status))
    os.environ.pop("TXQ_HIBF_LAYOUT_ORDER", None)
    same = np.array_equal(masks["layout"][0], masks["user"][0]) and masks["layout"][1] == masks["user"][1]
    if not same:
        raise SystemExit("layout-order and user-order query masks differ")
    # the same batch on SUB-TREE shards of the tree (all on this GPU; on a multi-GPU node shard r lives on device r): one expansion
    # drives all shards (txe_query_masks_sharded), the shards' full-width masks are ORed
    by_shards = {}
    for R in (2, 8):
        shards = [capi.Index.upload_hibf(user_bins, descs, shard_rank=r, n_shards=R, subtrees=True) for r in range(R)]
        capi.query_masks_sharded(shards, motifs[:20], False, 4)
        best = None
        for _ in range(2):
            t = time.perf_counter()
            m, status, stats = capi.query_masks_sharded(shards, motifs, False, 4)
            dt = time.perf_counter() - t
            if best is None or dt < best[0]:
                best = (dt, stats)
        if not (np.array_equal(m, masks["layout"][0]) and list(status) == masks["layout"][1]):
            raise SystemExit("sub-tree shards give other masks than the unsharded tree")
        by_shards[str(R)] = {"seconds": best[0], "queries_per_s": len(motifs) / best[0], "ibfs_per_shard": [int(s_.info.n_ibf) for s_ in shards],
                             "stages": best[1]["stages"], "ops": best[1]["ops"], "dense_ops": best[1]["dense_ops"]}
        for s_ in shards:
            s_.free()
    ix.free()
    return {
        "tree": "general HIBF, %d user bins, tmax %d: %d IBFs" % (user_bins, tmax, len(descs)), "kmers": n, "build_s": round(build_s, 1),
        "tree_bytes": int(info.device_bytes), "mask_words_user_order": int(W),
        "probe_user_order": {"what": "txq_probe_device (descent kernels, masks in user-bin order)", "seconds": t_user, "kmers_per_s": n / t_user,
                             "mask_GBps": n * W * 8 / t_user / 1e9},
        "probe_layout_order": {"what": "one session stage: k-mer table upload + layout-order rows of all k-mers + one op + result (txq_run_programs)",
                               "seconds": t_layout, "kmers_per_s": n / t_layout},
        "queries_layout_order": timings["layout"], "queries_user_order": timings["user"], "query_masks_identical": same,
        "queries_layout_order_sub_tree_shards_on_this_gpu": by_shards}


def main():
    import torch
    from tetrex_amd import capi
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    tmax = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    user_bins = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
    capi.init(0)
    print(json.dumps(measure(capi, torch, n, tmax, user_bins)))


if __name__ == "__main__":
    main()
