#!/usr/bin/env python3
"""bench.py — k-mer IBF probes/s of the MI355X TetRex probe path (BASELINE.json metric).

One "step" = one pass of the bulk_contains hot path (k-mer -> h hashes -> gather h bin-wide
rows -> AND -> per-bin hit mask written to HBM) over one resident batch of synthetic k-mers.

Workload (config.workload = "S-IBF-1024", BASELINE configs[1] shape):
  1024 bins per GPU, h = 3, m = compute_bitcount(200000, 0.05f) = 1,247,045 rows (159.6 MB),
  filled by inserting 200,000 uniform 20-bit values (k = 4 x 5 bits/residue) per bin with the
  real hash; probe batch = 2^24 uniform 20-bit k-mers (splitmix64, fixed seeds).
Multi-GPU (--gpus N, launched by torch.distributed.run), two modes:
  --scaling weak (default): the index has 1024*N bins and is sharded by bin-word columns, 1024 bins per rank
      (BASELINE configs[3] layout at N = 8); every rank probes the same batch against its own column shard; no
      collective on the probe path (bins are independent).  value = N * k-mers / time in SHARD-probes/s (one k-mer
      against one 1024-bin shard); at N = 1 a shard-probe is a probe of the whole index.
  --scaling strong: the FIXED 8192-bin x 62.5 M-row index (S-IBF-8192, 64 GB) cut into N column shards; value =
      k-mers / time = probes of the whole index per second; bytes_per_probe is the per-GPU share (4104 B at N = 1,
      520 B at N = 8).
The default line also carries `roofline_hbm`: the same probe kernel on an 8 GB matrix (the 8192-bin config's
per-GPU shard), which does not fit the 256 MB Infinity Cache that the 160 MB S-IBF-1024 matrix lives in.

Output: ONE JSON line on rank 0 (contract in the task description), including
  roofline     — algorithmic bytes of the probe kernel / its HIP-event duration vs 8 TB/s HBM,
  cpu_baseline — the CPU oracle (oracle/, a port of the reference path) timed on this host.
Two further legs report the other BASELINE figures without touching `value`:
  end_to_end   — queries/s, regex -> candidate-bin mask, on the same index, with its own cpu_baseline
                 (the oracle's single-threaded query(), ~10 s, masks compared bit for bit); at N = 1 also
                 end_to_end.hibf_batch: the same motif batch on a 1024-user-bin HIBF (BASELINE configs[2]);
  hibf         — k-mers/s of the HIBF descent on a 65536-user-bin tree (BASELINE configs[4] shape);
  hibf_1024    — the same on a 1024-user-bin tree (the Swissprot-HIBF shape of BASELINE configs[2]).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def usable_cpus():
    """CPUs this process (and its sibling ranks) may use: affinity mask and the container's CPU quota (cgroup v2 cpu.max,
    v1 cpu.cfs_quota_us) — whichever is smaller."""
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cpus = min(cpus, max(1, round(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                cpus = min(cpus, max(1, round(q / p)))
        except (OSError, ValueError):
            pass
    return cpus


def self_launch(n_gpus):
    """Run this very command under `python -m torch.distributed.run --nproc-per-node N` (one rank per GPU, rendezvous on
    127.0.0.1 at a free port) as a child process; stdout / stderr pass through, the child's exit code is returned."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def splitmix64(seed, n, start=0):
    with np.errstate(over="ignore"):
        x = np.uint64(seed) + (np.arange(1, n + 1, dtype=np.uint64) + np.uint64(start)) * np.uint64(0x9E3779B97F4A7C15)
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def compute_bitcount(n, fpr):
    """Rows of an IBF sized like the reference does (include/index_ibf.h:133-139):
    ceil(-n ln p / ln^2 2) with ln p evaluated in float."""
    import math
    return int(math.ceil(-float(n) * float(np.log(np.float32(fpr))) / (math.log(2.0) ** 2)))


def workload_name(bins_local, m, args):
    """SURVEY.md §8(d) names: S-IBF-1024 (the default), S-IBF-8192 (8192 bins x 62.5 M rows = 64 GB on
    one GPU, BASELINE configs[3] unsharded); anything else spells its shape out."""
    if bins_local == 1024 and args.rows == 0:
        return "S-IBF-1024"
    if bins_local == 8192 and m == 62500000:
        return "S-IBF-8192"
    return "S-IBF-%d-rows%d" % (bins_local, m)


def build_index(capi, torch, bins_total, bins_local, m, h, rank, world, per_bin, value_bits):
    """Device-side construction: per_bin uniform values into each of this rank's bins."""
    ix = capi.Index.create_ibf(bins_total, m, h, shard_rank=rank, n_shards=world)
    first_bin = int(ix.info.shard_word0) * 64
    chunk_bins = 64
    shift = np.uint64(64 - value_bits)
    for b0 in range(0, bins_local, chunk_bins):
        nb = min(chunk_bins, bins_local - b0)
        n = nb * per_bin
        # value stream is a function of the GLOBAL bin id, so a shard holds the same bits at every N
        vals = splitmix64(1, n, start=(first_bin + b0) * per_bin) >> shift
        bins_of = (first_bin + b0 + np.repeat(np.arange(nb, dtype=np.uint32), per_bin)).astype(np.uint32)
        dv = torch.from_numpy(vals.view(np.int64)).cuda()
        db = torch.from_numpy(bins_of.view(np.int32)).cuda()
        ix.emplace_device(dv.data_ptr(), db.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    return ix


def pmc_traffic(tag, n, W, h):
    """HBM bytes per probe-kernel launch from the committed rocprofv3 PMC passes
    (profiles/r*_pmc_traffic_<tag>.json: 2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes, gfx950 correction).
    PMC counters cannot be collected from inside this process: the figure comes from SEPARATE passes of this
    very command (tools/pmc_traffic.sh), is only reported for the exact workload those passes ran, and is
    labelled with the file it was read from.  Returns (bytes or None, source or None)."""
    if not (n == (1 << 24) and W == 16 and h == 3):
        return None, None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_%s.json" % tag)))
    if not files:
        return None, None
    with open(files[-1]) as f:
        kernels = json.load(f)["kernels"]
    for name, k in kernels.items():
        if name.startswith("void txq::probe_kernel<8, 3") and "hbm_traffic_bytes_per_launch_corrected" in k:
            return k["hbm_traffic_bytes_per_launch_corrected"], "profiles/" + os.path.basename(files[-1]) + " (separate rocprofv3 --pmc passes, not measured in this run)"
    return None, None


def timed_probe_steps(torch, dist, ix, d_kmers, n, d_masks, stream, steps, warmup, world, coll_device):
    """W untimed + K timed launches of the probe path, barrier + synchronize on both sides, max over ranks.
    Returns (elapsed seconds, mean kernel seconds by HIP events on the launch stream)."""
    def step():
        ix.probe_device(d_kmers.data_ptr(), n, d_masks.data_ptr(), None, stream.cuda_stream)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, float(np.mean([a.elapsed_time(b) for a, b in evs])) / 1e3


def hbm_leg(capi, torch, args, h):
    """The out-of-cache leg of the default line: the SAME probe kernel on the 8192-bin config's per-GPU shard — 1024
    bins x 62.5 M rows = 8 GB, 30x the Infinity Cache — with uniform 40-bit k-mers, so every row gather goes to HBM.
    Rank-local, after the timed region; does not touch `value`."""
    rows, bits, per_bin = 62500000, 40, 20000
    ix = build_index(capi, torch, 1024, 1024, rows, h, 0, 1, per_bin, bits)
    W = ix.shard_words
    n = args.kmers
    kmers = splitmix64(12, n) >> np.uint64(64 - bits)
    d_kmers = torch.from_numpy(kmers.view(np.int64)).cuda()
    d_masks = torch.empty((n, W), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream()
    steps = max(5, min(args.steps, 20))
    elapsed, kernel_s = timed_probe_steps(torch, None, ix, d_kmers, n, d_masks, stream, steps, 2, 1, "cuda")
    # parity: a slice of the timed output against single probes of the same device matrix (no oracle copy of 8 GB)
    present = build_probe_check(ix, d_masks, kmers, torch)
    bytes_per_probe = h * W * 8 + W * 8 + 8
    achieved = bytes_per_probe * n / kernel_s / 1e9
    traffic, source = pmc_traffic("S-IBF-1024-rows62500000", n, W, h)
    out = {"bound": "hbm", "kernel": "txq::probe_kernel<8,3,2,false,NoRoot>", "workload": "S-IBF-1024-rows62500000 (8 GB, the 8192-bin index's per-GPU shard)",
           "matrix_bytes": int(ix.info.device_bytes), "kmer_bits": bits, "kmers_per_step": n, "steps": steps,
           "probes_per_s": n * steps / elapsed, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "bytes_per_probe": bytes_per_probe, "avg_kernel_ms": kernel_s * 1e3, "traffic": traffic, "self_check_rows": present}
    if source:
        out["traffic_source"] = source
    ix.free()
    return out


def build_probe_check(ix, d_masks, kmers, torch):
    """Size-independent property on the big matrix: the batch kernel's masks equal those of the same k-mers probed
    one small batch at a time through the host-buffer entry point (different launch geometry, same matrix)."""
    sample = np.concatenate([kmers[:257], kmers[-129:]])
    want = ix.probe(sample)
    got = torch.cat([d_masks[:257], d_masks[-129:]]).cpu().numpy().view(np.uint64)
    if not np.array_equal(got, want):
        raise SystemExit("bench: batch probe and single probes disagree on the out-of-cache matrix")
    return int(sample.size)


def cpu_baseline(ix, m, h, bins_local, kmers, sample, threads):
    """The oracle (CPU restatement of the reference probe path) on this host's cores.
    Uses the device-built matrix so GPU and CPU probe the same bits."""
    import oracle as O
    words = ix.download_words_rows(m)
    ox = O.Index.ibf(bins_local, m, h, dna=False, k=4)
    ox.set_words(words)
    q = kmers[:sample]
    ox.probe(q[:4096])  # touch code and pages
    t0 = time.perf_counter()
    out = ox.probe(q, threads=threads)
    dt = time.perf_counter() - t0
    return out, q.size / dt, dt


def oracle_check_rest(index, meta, motifs, masks, which, budget_s, label):
    """Parity beyond the timed baseline: the masks of the queries `which` (indices into motifs) against the CPU oracle, in a child
    process (oracle/check_masks.py, no GPU in it) on this rank's share of the host threads, stopped at the deadline — the oracle
    enumerates every state, a motif that begins with wildcards can cost it minutes.  index: {"words": ...} (flat IBF) or a list
    of IBF descriptors (HIBF).  A mismatch ends the bench."""
    import shutil
    import subprocess
    import tempfile
    import threading
    which = list(which)
    if not which:
        return {"masks_compared": 0, "unfinished": 0}
    threads = max(1, min(32, usable_cpus()))
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    work = tempfile.mkdtemp(prefix="tetrex_oracle_", dir=base)
    try:
        json.dump(dict(meta, threads=threads, motifs=[motifs[i] for i in which]), open(os.path.join(work, "meta.json"), "w"))
        if isinstance(index, dict):
            np.savez(os.path.join(work, "index.npz"), words=index["words"])
        else:
            arrays = {"n": np.int64(len(index))}
            for i, d in enumerate(index):
                arrays.update({"bins_%d" % i: np.int64(d["bins"]), "rows_%d" % i: np.int64(d["bin_size"]), "h_%d" % i: np.int64(d["hash_funs"]),
                               "words_%d" % i: d["words"], "next_%d" % i: d["next_ibf_id"], "user_%d" % i: d["tb_to_user"]})
            np.savez(os.path.join(work, "index.npz"), **arrays)
        np.save(os.path.join(work, "masks.npy"), np.ascontiguousarray(masks[which]))
        t0 = time.perf_counter()
        child = subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "check_masks.py"), work], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        lines = []
        reader = threading.Thread(target=lambda: lines.extend(child.stdout), daemon=True)
        reader.start()
        reader.join(budget_s)
        stopped = reader.is_alive()
        if stopped:
            child.kill()  # (exactly the child started here)
            reader.join(10)
        child.wait()
        got = [ln.split() for ln in list(lines) if ln.strip()]
        bad = [motifs[which[int(x[1])]] for x in got if x[0] == "MISMATCH"]
        if bad:
            raise SystemExit("bench: %s: the candidate-bin mask of %r differs from the CPU oracle" % (label, bad[0]))
        if not stopped and child.returncode != 0:
            return {"error": "oracle/check_masks.py failed: " + child.stderr.read()[-300:]}
        ok = sum(1 for x in got if x[0] == "ok")
        return {"masks_compared": ok, "refused_by_the_oracle": sum(1 for x in got if x[0] == "refused"), "unfinished": len(which) - len(got),
                "threads": threads, "seconds": time.perf_counter() - t0,
                **({"stopped_at_the_deadline": True} if stopped else {})}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def cpu_query_baseline(ix, m, h, bins_local, k, motifs, gpu_masks, budget_s):
    """cpu_baseline of the end-to-end leg: the oracle's query() (restatement of preprocess ->
    construct_kgraph -> OTFCollector::collect with immediate pruning, one thread like the reference) on
    the first motifs of the same batch, for about `budget_s` seconds, on the device-built matrix.  The
    masks double as a parity check of the GPU result."""
    import oracle as O
    ox = O.Index.ibf(bins_local, m, h, dna=False, k=k)
    ox.set_words(ix.download_words_rows(m))
    done, compared, quirky, extra, t0 = 0, 0, 0, 0.0, time.perf_counter()
    for i, rx in enumerate(motifs):
        try:
            mask, st = ox.query(rx, with_stats=True)
        except Exception:  # noqa: BLE001 - a motif the reference path cannot search either
            done += 1
            continue
        done += 1
        if st["quirk_merges"]:  # the reference merges states of different length there (implementation-defined result):
            t_extra = time.perf_counter()  # compare with the oracle under well-defined merges; not part of the baseline's time
            mask = ox.query(rx, well_defined=True)
            extra += time.perf_counter() - t_extra
            quirky += 1
        if not np.array_equal(mask, gpu_masks[i]):
            raise SystemExit("bench: the candidate-bin mask of %r differs from the CPU oracle" % rx)
        compared += 1
        if time.perf_counter() - t0 - extra > budget_s:
            break
    dt = time.perf_counter() - t0 - extra
    # ... and the REST of the batch, outside the baseline's time: every mask of the timed batch is checked by the oracle
    rest = oracle_check_rest({"words": ox.words()}, {"kind": "ibf", "bins": bins_local, "rows": m, "h": h, "dna": False, "k": k}, motifs, gpu_masks,
                             range(done, len(motifs)), 3 * budget_s, "end-to-end batch")
    return {"value": done / dt, "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": "first %d motifs of the same batch, single thread, %.1f s" % (done, dt), "masks_compared": compared,
            "compared_under_well_defined_merges": quirky, "rest_of_the_batch": rest,
            "masks_compared_in_all": compared + rest.get("masks_compared", 0)}


def end_to_end_queries(ix, torch, dist, world, rank, args):
    """Second headline metric (BASELINE.json): end-to-end queries/s = regex -> candidate-bin mask.
    Host: C++ front-end (regex -> k-graph -> staged frontier expansion); device: probe + mask-DAG
    executor with dead-state feedback; N>1: RCCL all-gather of the final per-query masks.
    Verification of the candidate bins (disk + regex scan) is not part of this figure.
    Runs after the timed probe steps; it does not touch `value`."""
    from motifs import random_prosite_motifs
    from tetrex_amd.dist import gather_final_masks
    k = max(2, args.kmer_bits // 5)  # peptide k-mers, 5 bits per residue (4 for the default 20-bit values)
    single = "LMA(E|Q)GLYN"  # BASELINE configs[1] motif
    motifs = random_prosite_motifs(args.motifs, 6)
    # a batch without gaps/wildcards (literal residues and residue classes only) for contrast
    plain = random_prosite_motifs(args.motifs, 7, wildcard=0.0, classes=0.3, ranges=0.0)
    err = None
    local = {}
    try:  # rank-local work first: a failure here must not leave the other ranks waiting in a collective
        ix.query_masks([single], False, k)  # warm-up (library load, first launches)
        lat = []
        for _ in range(20):
            t0 = time.perf_counter()
            ix.query_masks([single], False, k)
            lat.append(time.perf_counter() - t0)
        ix.query_masks(random_prosite_motifs(args.motifs, 9, wildcard=0.0, classes=0.3, ranges=0.0), False, k)  # warm, as below
        tp = time.perf_counter()
        _, plain_status, plain_stats = ix.query_masks(plain, False, k)
        plain_s = time.perf_counter() - tp
        # the timed batch runs warm, like the timed probe steps: one batch of the same size and mix (another seed) first —
        # a session's slot arena, staging sets and scratch are allocated by the first batch that needs them
        warm = random_prosite_motifs(args.motifs, 8)
        tw = time.perf_counter()
        ix.query_masks(warm, False, k)
        local = {"lat": lat, "plain_status": plain_status, "plain_stats": plain_stats, "plain_s": plain_s, "first_batch_s": time.perf_counter() - tw}
    except Exception as e:  # noqa: BLE001 - reported in the JSON line
        err = repr(e)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    masks = status = stats = None
    repeats = []
    if err is None:
        try:
            masks, status, stats = ix.query_masks(motifs, False, k)
        except Exception as e:  # noqa: BLE001
            err = repr(e)
    t1 = time.perf_counter()
    if err is None and world == 1:
        # one GPU: the batch is timed three times and the best run counts, like the other legs (a single run is at the mercy of
        # whatever else the host's sixteen CPUs were given to do in those 9 ms); all three are reported
        repeats = [t1 - t0]
        for _ in range(2):
            ta = time.perf_counter()
            m2, s2, st2 = ix.query_masks(motifs, False, k)
            tb = time.perf_counter()
            repeats.append(tb - ta)
            if tb - ta < t1 - t0:
                t0, t1, masks, status, stats = ta, tb, m2, s2, st2
    gather_s = 0.0
    total = t1 - t0
    if world > 1:
        ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=args.coll_device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            err = err or "another rank failed"
        else:
            loc = torch.from_numpy(masks.view(np.int64)).to(args.coll_device)
            full = gather_final_masks(loc, ix.info.mask_words)
            torch.cuda.synchronize()
            gather_s = time.perf_counter() - t1
            t = torch.tensor([t1 - t0 + gather_s], dtype=torch.float64, device=args.coll_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            total = float(t.item())
            assert full.shape == (len(motifs), int(ix.info.mask_words))
    if err is not None:
        return {"error": err}
    lat, plain_status, plain_stats, plain_s = local["lat"], local["plain_status"], local["plain_stats"], local["plain_s"]
    enumerated = None
    if world == 1:
        # ALL masks of the timed batch against the same batch with every state enumerated (TETREX_DENSE=0: no blocks, no table of
        # k-mer masks — ordinary ops on probed masks only); the CPU oracle below covers as many motifs as its 10 s allow
        knob = os.environ.get("TETREX_DENSE")
        os.environ["TETREX_DENSE"] = "0"
        try:
            ta = time.perf_counter()
            ref_masks, ref_status, ref_stats = ix.query_masks(motifs, False, k)
            ref_dt = time.perf_counter() - ta
        finally:
            if knob is None:
                del os.environ["TETREX_DENSE"]
            else:
                os.environ["TETREX_DENSE"] = knob
        if not (np.array_equal(masks, ref_masks) and list(status) == list(ref_status)):
            raise SystemExit("bench: the end-to-end batch gives other masks with dense blocks than with enumerated states")
        enumerated = {"what": "the same batch with TETREX_DENSE=0 (no blocks: states enumerated and pruned through host feedback)",
                      "seconds": ref_dt, "ops": ref_stats["ops"], "masks_identical": True}
    big = None
    if world == 1 and not args.no_big_batch:
        # BASELINE configs[3]'s batch size: 10 000 motifs of the same mix in ONE call (one warm run, best of three timed)
        try:
            many = random_prosite_motifs(10 * args.motifs, 6)
            ix.query_masks(many, False, k)
            runs = []
            for _ in range(3):
                ta = time.perf_counter()
                masks_many, st_many, stats_many = ix.query_masks(many, False, k)
                runs.append((time.perf_counter() - ta, stats_many, st_many, masks_many))
            best = min(runs, key=lambda r: r[0])
            if many[:len(motifs)] == motifs and not np.array_equal(best[3][:len(motifs)], masks):  # (its first motifs are the batch above)
                raise SystemExit("bench: the first motifs of the 10x batch give other masks than the same motifs as a batch of their own")
            big = {"motifs": len(many), "k": k, "seconds": best[0], "queries_per_s": len(many) / best[0], "timed_runs_seconds": [r[0] for r in runs],
                   "first_masks_equal_the_batch_above": bool(many[:len(motifs)] == motifs),
                   "refused_fraction": float(sum(1 for s_ in best[2] if s_)) / len(many), **best[1]}
        except Exception as e:  # noqa: BLE001
            big = {"error": repr(e)}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_query_baseline(ix, args.m_rows, args.hash, args.bins_per_gpu, k, motifs, masks, args.cpu_query_seconds)
    return {
        **({"cpu_baseline": cpu} if cpu else {}),
        "metric": "end-to-end queries/sec (regex -> candidate-bin mask, verification excluded)",
        "batch_queries_per_s": len(motifs) / total,
        **({"collective": {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                           "op": "all_gather of the final masks (%d x %d words per rank)" % (len(motifs), int(ix.shard_words))}} if world > 1 else {}),
        "batch": {"motifs": len(motifs), "seconds": total, "gather_seconds": gather_s, "failed": int(sum(1 for s in status if s)),
                  "warmup": "one batch of the same size and mix from another seed", "first_batch_seconds": local["first_batch_s"],
                  **({"timed_runs_seconds": repeats, "seconds_is": "the best of the timed runs"} if repeats else {}),
                  "k": k, "refused_fraction": float(sum(1 for s_ in status if s_)) / len(motifs), **stats, "mean_candidate_bins": float(np.unpackbits(masks.view(np.uint8), axis=1).sum(axis=1).mean()),
                  **({"enumerated_states": enumerated} if enumerated else {})},
        **({"batch_10x": big} if big else {}),
        "batch_no_wildcards": {"motifs": len(plain), "seconds": plain_s, "queries_per_s": len(plain) / plain_s,
                               "failed": int(sum(1 for s in plain_status if s)), **plain_stats},
        "single_query": {"motif": single, "median_latency_ms": float(np.median(lat)) * 1e3,
                         "queries_per_s": 1.0 / float(np.median(lat))},
    }


def hibf_end_to_end(capi, torch, args):
    """BASELINE configs[2]: the end-to-end batch (1000 PROSITE-style motifs, k = 4) on a 1024-user-bin peptide HIBF — 16
    children of 64 bins, h = 3, 20 000 values per bin, every IBF filled on the device with the real hash.  Its dense steps
    read the index's table of all 4-mers' masks (csrc/txq_exec.hip ensure_kmer_table / TableRows: membership_for of every
    packed value, built with the tree's descent when the first batch arrives); `interleaved_rows_seconds` is the same batch
    with the table off (TXQ_KMER_TABLE_MB=0: steps fused on the tree's interleaved children, InterleavedRows).  Check inside
    this run: the same batch once more with the steps sent through the generic HIBF descent (TXQ_DENSE_TREE=0: k-mers
    written out, hibf_probe, combine — the path the parity tests pin to the oracle); all masks must be identical.  N = 1 only;
    does not touch `value`."""
    from motifs import random_prosite_motifs
    user_bins, children, per_bin, h = 1024, 16, 20000, 3
    per_child = user_bins // children
    rng = np.random.default_rng(5)

    def filled(bins, rows, vals, bins_of):
        ix = capi.Index.create_ibf(bins, rows, h)
        dv = torch.from_numpy(vals.view(np.int64)).cuda()
        db = torch.from_numpy(bins_of.astype(np.uint32).view(np.int32)).cuda()
        ix.emplace_device(dv.data_ptr(), db.data_ptr(), vals.size, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        words = ix.download_words_rows(rows)
        ix.free()
        return words

    values = [rng.integers(0, 1 << 20, size=per_bin, dtype=np.uint64) for _ in range(user_bins)]
    m_child = compute_bitcount(per_bin, 0.05)
    m_root = compute_bitcount(per_bin * per_child, 0.05)
    descs, rv, rb = [None], [], []
    for c in range(children):
        v = np.concatenate(values[c * per_child:(c + 1) * per_child])
        descs.append(dict(bins=per_child, bin_size=m_child, hash_funs=h,
                          words=filled(per_child, m_child, v, np.repeat(np.arange(per_child, dtype=np.uint32), per_bin)),
                          next_ibf_id=np.zeros(per_child, dtype=np.uint64), tb_to_user=np.arange(c * per_child, (c + 1) * per_child, dtype=np.uint64)))
        rv.append(v)
        rb.append(np.full(v.size, c, dtype=np.uint32))
    descs[0] = dict(bins=children, bin_size=m_root, hash_funs=h, words=filled(children, m_root, np.concatenate(rv), np.concatenate(rb)),
                    next_ibf_id=np.arange(1, children + 1, dtype=np.uint64), tb_to_user=np.full(children, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64))
    ix = capi.Index.upload_hibf(user_bins, descs)
    motifs = random_prosite_motifs(args.motifs, 6)
    ix.query_masks(motifs[:10], False, 4)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        masks, status, stats = ix.query_masks(motifs, False, 4)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, stats, masks, status)
    knob = os.environ.get("TXQ_KMER_TABLE_MB")
    os.environ["TXQ_KMER_TABLE_MB"] = "0"
    try:
        rows_dt = None
        for _ in range(2):
            t0 = time.perf_counter()
            rows_masks, _, _ = ix.query_masks(motifs, False, 4)
            dt = time.perf_counter() - t0
            rows_dt = dt if rows_dt is None else min(rows_dt, dt)
    finally:
        if knob is None:
            del os.environ["TXQ_KMER_TABLE_MB"]
        else:
            os.environ["TXQ_KMER_TABLE_MB"] = knob
    if not np.array_equal(best[2], rows_masks):
        raise SystemExit("bench: the HIBF batch gives other masks through the table of all k-mers' masks than with rows gathered from the tree")
    knob = os.environ.get("TXQ_DENSE_TREE")
    os.environ["TXQ_DENSE_TREE"] = "0"
    try:
        t0 = time.perf_counter()
        ref_masks, ref_status, _ = ix.query_masks(motifs, False, 4)
        ref_dt = time.perf_counter() - t0
    finally:
        if knob is None:
            del os.environ["TXQ_DENSE_TREE"]
        else:
            os.environ["TXQ_DENSE_TREE"] = knob
    ix.free()
    if not (np.array_equal(best[2], ref_masks) and list(best[3]) == list(ref_status)):
        raise SystemExit("bench: the HIBF batch gives other masks with fused dense steps than through the generic HIBF descent")
    # ... and against the CPU oracle's HIBF (membership_for restated, oracle/txo_ibf.hpp): every motif of the batch
    rest = oracle_check_rest(descs, {"kind": "hibf", "bins": user_bins, "dna": False, "k": 4}, motifs, best[2], [i for i in range(len(motifs)) if not best[3][i]],
                             2 * args.cpu_query_seconds, "HIBF batch")
    compared = rest.get("masks_compared", 0)
    return {"workload": "BASELINE configs[2]: %d PROSITE-style motifs on a 1024-user-bin HIBF (16 x 64 bins, k=4, h=3, %d values per bin)" % (len(motifs), per_bin),
            "k": 4, "oracle_masks_compared": compared, "cpu_oracle": rest, "refused_fraction": float(sum(1 for x in best[3] if x)) / len(motifs),
            "queries_per_s": len(motifs) / best[0], "seconds": best[0], "failed": int(sum(1 for x in best[3] if x)),
            "mean_candidate_bins": float(np.unpackbits(best[2].view(np.uint8), axis=1).sum(axis=1).mean()), **best[1],
            "interleaved_rows_seconds": rows_dt,
            "checked_against": "the same batch with its dense steps through the generic HIBF descent (TXQ_DENSE_TREE=0), and with rows gathered from the tree (TXQ_KMER_TABLE_MB=0)",
            "masks_identical": bool(np.array_equal(best[2], ref_masks) and list(best[3]) == list(ref_status)),
            "generic_descent_seconds": ref_dt}


def k6_end_to_end(capi, torch, args, check=True):
    """The reference's DEFAULT k (include/arg_parse.h:12: k = 6; the README's Swissprot scenario, README.md:84-109): a
    1024-bin flat IBF over Swissprot-SHAPED content — per bin the 6-mers of 200 000 uniform random residues, inserted on
    the device with the real hash (h = 3, rows for fpr 0.05) — and a batch of 200 PROSITE-style motifs with wildcards and
    x(m,n) gaps (the batch of tests/perf_cli_swissprot_shape.py).  At k = 6 a block of suffixes has 21^5 entries of which
    such an index keeps a few alive: the expansion keeps wildcard lists as TRACKED blocks (include/txq_program.h: live
    lists, steps pushed from the live entries, blocks laid out inside their lists' geometries).  Checks inside this run:
    the same batch with dense blocks switched off (TETREX_DENSE=0: every state enumerated and pruned by the host) must give
    identical masks, and the CPU oracle is compared on the motifs it answers within its budget.  N = 1 only."""
    from motifs import random_prosite_motifs
    import oracle as O
    bins, per_bin, h, k = 1024, 200000, 3, 6
    m = compute_bitcount(per_bin, 0.05)
    base_code = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19], dtype=np.uint64)  # "ACDEFGHIKLMNPQRSTVWY" in the Base alphabet

    def build(residues, seed):
        """bins x m rows, h hashes: per bin the k-mers of `residues` uniform random residues, inserted on the device"""
        index = capi.Index.create_ibf(bins, m, h)
        rng = np.random.default_rng(seed)
        for b0 in range(0, bins, 64):  # 64 bins at a time: 12.8 M values
            nb = min(64, bins - b0)
            codes = base_code[rng.integers(0, 20, size=(nb, residues))]
            vals = np.zeros((nb, residues - k + 1), dtype=np.uint64)
            for j in range(k):
                vals = (vals << np.uint64(5)) | codes[:, j:residues - k + 1 + j]
            bins_of = np.repeat(np.arange(b0, b0 + nb, dtype=np.uint32), vals.shape[1])
            dv = torch.from_numpy(vals.reshape(-1).view(np.int64)).cuda()
            db = torch.from_numpy(bins_of.view(np.int32)).cuda()
            index.emplace_device(dv.data_ptr(), db.data_ptr(), vals.size, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        return index

    ix = build(per_bin, 11)
    motifs = random_prosite_motifs(200, 3, wildcard=0.05, ranges=0.02, min_len=8, max_len=14)
    ix.query_masks(random_prosite_motifs(200, 4, wildcard=0.05, ranges=0.02, min_len=8, max_len=14), False, k)  # warm: arena, staging sets
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        masks, status, stats = ix.query_masks(motifs, False, k)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, stats, masks, status)
    if not check:  # (profiling runs: the timed batches only)
        ix.free()
        return {"k": k, "seconds": best[0], **best[1]}
    knob = os.environ.get("TETREX_DENSE")
    os.environ["TETREX_DENSE"] = "0"
    try:
        t0 = time.perf_counter()
        ref_masks, ref_status, ref_stats = ix.query_masks(motifs, False, k)
        ref_dt = time.perf_counter() - t0
    finally:
        if knob is None:
            del os.environ["TETREX_DENSE"]
        else:
            os.environ["TETREX_DENSE"] = knob
    if not (np.array_equal(best[2], ref_masks) and list(best[3]) == list(ref_status)):
        raise SystemExit("bench: the k = 6 batch gives other masks with tracked blocks than with enumerated states")
    # CPU oracle on the motifs whose leading residues are fixed (it enumerates every state: a wildcard in front costs it minutes)
    ox = O.Index.ibf(bins, m, h, dna=False, k=k)
    ox.set_words(ix.download_words_rows(m))
    compared, checked, t0 = 0, set(), time.perf_counter()
    for i, rx in enumerate(motifs):
        if "." in rx[:10]:
            continue
        checked.add(i)
        try:
            want = ox.expected_mask(rx)[0]
        except Exception:  # noqa: BLE001 - a motif the reference path cannot search either
            continue
        if not np.array_equal(want, best[2][i]):
            raise SystemExit("bench: the candidate-bin mask of %r (k = 6) differs from the CPU oracle" % rx)
        compared += 1
        if time.perf_counter() - t0 > 5.0:
            break
    cpu_dt = time.perf_counter() - t0
    # ... and every other motif of the batch — above all the ones that begin with wildcards, which tracked blocks exist for — in
    # a child process on all host threads, until its deadline
    others = [i for i in range(len(motifs)) if i not in checked and not best[3][i]]
    rest = oracle_check_rest({"words": ox.words()}, {"kind": "ibf", "bins": bins, "rows": m, "h": h, "dna": False, "k": k}, motifs, best[2], others,
                             3 * getattr(args, "cpu_query_seconds", 10.0), "k = 6 batch")
    ix.free()
    # ... and the WHOLE batch once more on a thinner index of the same shape (same bins, rows and hashes, a fifth of the residues per
    # bin): there a k-mer is in ~1 bin instead of ~54, the oracle's states die out after a residue or two and it answers the motifs
    # that begin with wildcards too — the ones tracked blocks exist for
    thin_residues = per_bin // 5
    thin = build(thin_residues, 12)
    t_masks, t_status, t_stats = thin.query_masks(motifs, False, k)
    thin_words = thin.download_words_rows(m)
    thin.free()
    thin_rest = oracle_check_rest({"words": thin_words}, {"kind": "ibf", "bins": bins, "rows": m, "h": h, "dna": False, "k": k}, motifs, t_masks,
                                  [i for i in range(len(motifs)) if not t_status[i]], 3 * getattr(args, "cpu_query_seconds", 10.0), "k = 6 batch on the thin index")
    thin_rest.update({"index": "%d bins x %d rows, h = %d: the 6-mers of %d random residues per bin" % (bins, m, h, thin_residues),
                      "motifs_with_candidate_bins": int((t_masks != 0).any(axis=1).sum()),
                      "tracked_queries": t_stats.get("tracked_queries"), "dense_ops": t_stats.get("dense_ops"), "ops": t_stats.get("ops")})
    refused = int(sum(1 for x in best[3] if x))
    roof = None
    try:  # the kernel of this leg against the HBM roofline: from the committed rocprofv3 passes of this very workload (tools/pmc_sparse.py)
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_sparse_kernel.json")))
        if files:
            with open(files[-1]) as f:
                roof = dict(json.load(f)["roofline"], source="profiles/" + os.path.basename(files[-1]) + " (separate rocprofv3 --stats / --pmc passes, not measured in this run)")
    except Exception:  # noqa: BLE001
        roof = None
    return {**({"roofline": roof} if roof else {}),
            "workload": "%d PROSITE-style motifs (5 %% wildcards, 2 %% x(m,n)) at k = 6 on a 1024-bin flat IBF of Swissprot-shaped bins "
                        "(6-mers of %d random residues per bin, h = 3, %d rows)" % (len(motifs), per_bin, m),
            "k": k, "queries_per_s": len(motifs) / best[0], "seconds": best[0], "refused": refused, "refused_fraction": refused / len(motifs),
            "mean_candidate_bins": float(np.unpackbits(best[2].view(np.uint8), axis=1).sum(axis=1).mean()), **best[1],
            "enumerated_states": {"what": "the same batch with TETREX_DENSE=0 (no blocks: states enumerated and pruned through host feedback)",
                                  "seconds": ref_dt, "ops": ref_stats["ops"], "masks_identical": True},
            "cpu_oracle": {"masks_compared": compared, "seconds": cpu_dt, "sample": "motifs of the batch without a wildcard among their first residues",
                           "the_other_motifs": rest, "masks_compared_in_all": compared + rest.get("masks_compared", 0),
                           "whole_batch_on_a_thin_index_of_the_same_shape": thin_rest}}


def verified_end_to_end(args):
    """End-to-end queries/s INCLUDING verification, timed like the reference times a query: wall-clock from after the index is
    loaded to the last output byte (include/query.h:256,287-289) — candidate masks on the GPU, then the candidate bins'
    FASTA files read from local disk and searched with the motif (host/verify.cpp: iter_disk_search / verify_fasta_hit of
    src/query.cpp:194-315 with this project's linear-time matcher; OpenMP over the candidate bins like the reference).
    The README's scenario (README.md:84-109) on synthetic data of its shape: 1024 FASTA bins of 200 000 uniform random
    residues (360-residue records), `tetrex index -k 6 -i` (flat IBF, h = 3, fpr 0.05), 200 PROSITE-style motifs through
    `tetrex query -f` with -t 1 and -t 16.  cpu_baseline: the CPU oracle's mask stage on the same index file (the motifs it
    answers in its budget; it enumerates every state, so motifs with a wildcard among their first residues are left out)
    plus the same verification single-threaded.  N = 1 only; does not touch `value`."""
    import subprocess
    import tempfile
    import oracle as O
    from motifs import random_prosite_motifs
    from tetrex_amd import host as H
    tetrex = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "tetrex")
    bins, per_bin, k = 1024, 200000, 6
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    rng = np.random.default_rng(11)
    out = {}
    with tempfile.TemporaryDirectory(prefix="tetrex_bench_") as work:
        files = []
        for b in range(bins):
            seq = aa[rng.integers(0, 20, size=per_bin)]
            path = os.path.join(work, "bin%04d.fa" % b)
            with open(path, "wb") as f:
                for i, start in enumerate(range(0, per_bin, 360)):
                    f.write(b">sp|%04d_%d\n" % (b, i))
                    f.write(seq[start:start + 360].tobytes())
                    f.write(b"\n")
            files.append(path)
        motifs = random_prosite_motifs(200, 3, wildcard=0.05, ranges=0.02, min_len=8, max_len=14)
        with open(os.path.join(work, "motifs.tsv"), "w") as f:
            for i, m in enumerate(motifs):
                f.write("M%03d\t%s\n" % (i, m))
        r = subprocess.run([tetrex, "index", "-k", str(k), "-i", "sp", *files], capture_output=True, text=True, cwd=work)
        if r.returncode != 0:
            return {"error": "tetrex index failed: " + r.stderr[-500:]}
        if os.environ.get("BENCH_CLI_TRACE"):  # (tools/verify_leg.py: what a cold process spends its mask stage on)
            r = subprocess.run([tetrex, "query", "-S", "-f", "-t", "16", "sp.ibf", "motifs.tsv"], capture_output=True, text=True, cwd=work,
                               env=dict(os.environ, TXQ_TRACE="1", TETREX_TRACE="1"))
            sys.stderr.write("".join(ln + "\n" for ln in r.stderr.splitlines() if ln.startswith("[t") or ln.startswith("{")))
        if os.environ.get("BENCH_CLI_ROCPROF"):  # (tools/verify_leg.py: the device's timeline of a COLD process; the binary itself after `--`)
            subprocess.run(["rocprofv3", "--kernel-trace", *(["--hip-trace"] if os.environ.get("BENCH_CLI_ROCPROF_HIP") else []), "--stats", "--output-format", "csv", "-d", os.environ["BENCH_CLI_ROCPROF"], "-o", "cold",
                            "--", tetrex, "query", "-S", "-f", "-t", "16", "sp.ibf", "motifs.tsv"], capture_output=True, text=True, cwd=work,
                           env=dict(os.environ, TMPDIR="/tmp"))
        runs = {}
        for threads in (1, 16):
            best = None
            for _ in range(3):  # the later runs read the FASTA files from the page cache, like a server that has seen them before
                # (and a fresh process sometimes waits 0.1 s for the driver to hand out its first gigabytes of block memory)
                r = subprocess.run([tetrex, "query", "-S", "-f", "-t", str(threads), "sp.ibf", "motifs.tsv"], capture_output=True, text=True, cwd=work)
                if r.returncode != 0:
                    return {"error": "tetrex query failed: " + r.stderr[-500:]}
                stats = [json.loads(ln) for ln in r.stderr.splitlines() if ln.startswith("{")]
                if os.environ.get("TETREX_TRACE"):  # (tools/verify_leg.py under TETREX_TRACE: where the verification's time goes)
                    sys.stderr.write("".join(ln + "\n" for ln in r.stderr.splitlines() if ln.startswith("[tetrex] verify_batch")))
                whole = [x for x in stats if "batch_seconds" in x][-1]
                mask = [x for x in stats if "mask_seconds" in x][-1]
                if best is None or whole["batch_seconds"] < best[0]["batch_seconds"]:
                    best = (whole, mask)
            whole, mask = best
            hits = sum(1 for i in range(len(motifs)) if os.path.exists(os.path.join(work, "M%03d.tsv" % i)) and os.path.getsize(os.path.join(work, "M%03d.tsv" % i)) > 0)
            runs["threads_%d" % threads] = {"queries_per_s": len(motifs) / whole["batch_seconds"], "seconds": whole["batch_seconds"],
                                            "mask_seconds": mask["mask_seconds"], "verify_seconds": whole["verify_seconds"],
                                            "refused_fraction": whole["refused"] / len(motifs), "motifs_with_verified_matches": hits}
        # CPU baseline: the oracle's mask stage on the index file the CLI wrote, plus the single-threaded verification measured above
        img = H.IndexFile.load(os.path.join(work, "sp.ibf"))
        d = img.describe()
        ibf = d["ibfs"][0]
        ox = O.Index.ibf(int(ibf["bins"]), int(ibf["bin_size"]), int(ibf["hash_funs"]), dna=False, k=k)
        ox.set_words(img.words(0))
        done, t0 = 0, time.perf_counter()
        for rx in motifs:
            if "." in rx[:10]:
                continue
            try:
                ox.query(rx)
            except Exception:  # noqa: BLE001
                pass
            done += 1
            if time.perf_counter() - t0 > 5.0:
                break
        cpu_mask_per_query = (time.perf_counter() - t0) / max(done, 1)
        per_query = cpu_mask_per_query + runs["threads_1"]["verify_seconds"] / len(motifs)
        out = {"metric": "end-to-end queries/sec INCLUDING verification (regex -> candidate bins -> verified matches on disk)",
               "workload": "%d PROSITE-style motifs, k = %d, %d FASTA bins of %d residues on local disk (flat IBF written by `tetrex index -i`)" % (len(motifs), k, bins, per_bin),
               "k": k, **runs,
               "cpu_baseline": {"value": 1.0 / per_query, "unit": "queries/s", "cores": 1, "kind": "port",
                                "sample": "oracle mask stage on %d motifs of the batch (%.4f s per query) + the same verification with one thread (%.4f s per query)"
                                          % (done, cpu_mask_per_query, runs["threads_1"]["verify_seconds"] / len(motifs))}}
    return out


def one_process_n_devices(capi, torch, n_devices, world, build_shard, motifs, warm, dna, k, what):
    """The C++ product deployment of the sharded index: ONE process drives all N devices — txq_init(N, ids),
    shard r on device r, one frontier expansion feeding every shard's session (ShardedStageExecutor, host/device_index.cpp),
    the shards' masks joined on the host (txe_query_masks_sharded; the seam of run_collection / run_multiple_queries,
    reference include/query.h:250-290,329-346).  Runs on rank 0 alone after the distributed legs, while the other ranks
    wait on the HOST (a gloo group: no barrier kernel spins on the devices rank 0 is measuring); build_shard(r) builds
    shard r of the leg's index on the current device.  Reports the batch time and what the shards' stages took."""
    capi.init_devices(list(range(n_devices)))
    shards = []
    t0 = time.perf_counter()
    try:
        for r in range(world):
            torch.cuda.set_device(r % n_devices)
            shards.append(build_shard(r))
        build_s = time.perf_counter() - t0
        devices = [int(s.info.device) for s in shards]
        capi.query_masks_sharded(shards, warm, dna, k)  # warm, like the other legs
        best = None
        for _ in range(2):
            t1 = time.perf_counter()
            masks, status, stats = capi.query_masks_sharded(shards, motifs, dna, k)
            dt = time.perf_counter() - t1
            if best is None or dt < best[0]:
                best = (dt, stats, masks, status)
        # the same batch on shard 0 alone must give shard 0's columns of the joined masks
        part, _, _ = shards[0].query_masks(motifs, dna, k)
        w0, nw = int(shards[0].info.shard_word0), shards[0].shard_words
        if not np.array_equal(part, best[2][:, w0:w0 + nw]):
            raise RuntimeError("joined masks differ from shard 0's own run in its columns")
        return {"what": "one process, %d device(s): txe_query_masks_sharded (one expansion, %d shard sessions, host-side join) — %s" % (n_devices, world, what),
                "devices": devices, "motifs": len(motifs), "queries_per_s": len(motifs) / best[0], "seconds": best[0], "k": k,
                "refused_fraction": float(sum(1 for x in best[3] if x)) / len(motifs), **best[1], "index_build_s": round(build_s, 1),
                "mask_words": int(best[2].shape[1])}, best[2]
    finally:
        for s in shards:
            s.free()
        torch.cuda.set_device(0)


def rank0_alone(torch, dist, host_group, rank, fn):
    """fn() on rank 0 while every other rank waits on the host (gloo monitored_barrier with a long timeout; their devices
    idle and their caches emptied).  A failure of fn is returned as {"error": ...}: it must not cost the contract line."""
    import datetime
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    dist.monitored_barrier(group=host_group, timeout=datetime.timedelta(minutes=30))
    res = None
    if rank == 0:
        try:
            res = fn()
        except Exception as e:  # noqa: BLE001
            res = {"error": repr(e)}
    dist.monitored_barrier(group=host_group, timeout=datetime.timedelta(minutes=30))
    return res


DNA_LETTERS = np.frombuffer(b"ACTG", dtype=np.uint8)  # the reference's 2-bit code (include/nucleotide_decomposer.h): (c >> 1) & 3


def dna_group_codes(torch, group, seq_len):
    """The 2-bit codes of the 64 sequences of bin group `group` (bins 64*group .. 64*group+63), generated on the device from a
    seed that depends on the group only — every rank of every N sees the same sequences."""
    g = torch.Generator(device="cuda")
    g.manual_seed(0x5EED0000 + group)
    return torch.randint(0, 4, (64, seq_len), generator=g, device="cuda", dtype=torch.int64)


def dna_canonical_kmers(torch, codes, k):
    """Canonical k-mer values (min of forward and reverse complement, complement = code ^ 2) of every window of every row."""
    n = codes.shape[1] - k + 1
    fwd = torch.zeros((codes.shape[0], n), dtype=torch.int64, device=codes.device)
    rc = torch.zeros_like(fwd)
    for j in range(k):
        w = codes[:, j:j + n]
        fwd |= w << (2 * (k - 1 - j))
        rc |= (w ^ 2) << (2 * j)
    return torch.minimum(fwd, rc)


def dna_build_shard(capi, torch, bins_total, rows, h, k, seq_len, r, R):
    """Column shard r of R of the DNA index on the current device: bin b holds the canonical k-mers of sequence b."""
    ix = capi.Index.create_ibf(bins_total, rows, h, shard_rank=r, n_shards=R)
    w0, nw = int(ix.info.shard_word0), ix.shard_words
    stream = torch.cuda.current_stream().cuda_stream
    for g in range(w0, w0 + nw):
        codes = dna_group_codes(torch, g, seq_len)
        vals = dna_canonical_kmers(torch, codes, k).reshape(-1)
        bins_of = (g * 64 + torch.arange(64, device="cuda", dtype=torch.int32)).repeat_interleave(seq_len - k + 1)
        ix.emplace_device(vals.data_ptr(), bins_of.data_ptr(), vals.numel(), stream)
        torch.cuda.synchronize()
    return ix


def dna_motifs(rng, windows, n_motifs):
    """The motif generator of tests/perf_config4_queries.py: 24-32 nt windows of the indexed sequences with up to two positions
    turned into a wildcard, a class or an optional residue.  windows: uint8 letters [n_motifs, 32]."""
    motifs = []
    for i in range(n_motifs):
        L = int(rng.integers(24, 33))
        w = list(windows[i, :L].tobytes().decode())
        for p in rng.choice(np.arange(4, L - 4), size=int(rng.integers(0, 3)), replace=False):
            p = int(p)
            r = rng.random()
            if r < 0.4:
                w[p] = "."
            elif r < 0.8:
                w[p] = "[" + "".join(sorted(set(w[p] + "ACGT"[int(rng.integers(0, 4))]))) + "]"
            else:
                w[p] = w[p] + "?"
        motifs.append("".join(w))
    return motifs


def dna_batch_8192(capi, torch, dist, args, rank, world):
    """BASELINE configs[3]: the 8192-bin DNA IBF (62.5 M rows, h = 3: 64 GB) cut into N column shards, k = 16, and a batch of
    10 000 motifs in ONE call per rank (every rank expands the batch; the path's only exchange is the all-gather of the final
    masks over RCCL — tetrex_amd/dist.py; shards are disjoint columns, so the gather IS the OR-reduce).  Every bin holds the
    canonical 16-mers of a random 100 kb sequence (generated and inserted on the device); a motif is a window of one bin's
    sequence (its home) with a few degenerate positions, so it must be found there (checked on all motifs), and the masks of a
    64-bin column are compared with the CPU oracle, whose matrix is built on the host from the same sequences by the oracle's
    own encoder.  This leg is strong-scaled whatever --scaling says (the index is fixed).  Reference seam:
    include/query.h:250-290,329-346 (run_collection / run_multiple_queries)."""
    from tetrex_amd.dist import gather_final_masks
    bins_total, h, k, seq_len = 8192, 3, 16, args.dna_seq_len
    rows, n_motifs = args.dna_rows, args.dna_motifs
    if 128 % world:
        return {"skipped": "%d ranks do not divide the 128 mask words" % world}, None
    t0 = time.perf_counter()
    rng = np.random.default_rng(4)
    home = rng.integers(0, bins_total, size=n_motifs)
    at = rng.integers(0, seq_len - 32, size=n_motifs)
    windows = np.zeros((n_motifs, 32), dtype=np.uint8)
    check_word = 77  # the mask column the CPU oracle rebuilds (at N > 1 it is not rank 0's: the comparison crosses the gather)
    check_codes = None
    for g in range(bins_total // 64):  # the motifs' windows: every rank regenerates every group's sequences (milliseconds on the device)
        sel = np.nonzero(home // 64 == g)[0]
        if sel.size or (g == check_word and rank == 0):
            codes = dna_group_codes(torch, g, seq_len)
            if sel.size:
                idx = torch.from_numpy(at[sel]).cuda()[:, None] + torch.arange(32, device="cuda")[None, :]
                windows[sel] = DNA_LETTERS[codes[torch.from_numpy(home[sel] % 64).cuda()[:, None], idx].cpu().numpy()]
            if g == check_word and rank == 0:
                check_codes = codes.cpu().numpy()
    err = ix = None
    nw = 0
    try:  # (rank-local: a failure here must still let this rank enter the collectives below)
        ix = dna_build_shard(capi, torch, bins_total, rows, h, k, seq_len, rank, world)
        nw = ix.shard_words
    except Exception as e:  # noqa: BLE001 - reported in the line
        err = repr(e)
    build_s = time.perf_counter() - t0
    motifs = dna_motifs(rng, windows, n_motifs)
    masks = status = stats = None
    runs = []
    try:
        if err is None:
            ix.query_masks(motifs[:200], True, k)  # warm: arena, staging sets
    except Exception as e:  # noqa: BLE001
        err = repr(e)
    if world > 1:
        dist.barrier()
    try:
        for _ in range((3 if world == 1 else 1) if err is None else 0):
            ta = time.perf_counter()
            m_, s_, st_ = ix.query_masks(motifs, True, k)
            runs.append(time.perf_counter() - ta)
            if masks is None or runs[-1] <= min(runs):
                masks, status, stats = m_, s_, st_
    except Exception as e:  # noqa: BLE001 - reported in the line
        err = repr(e)
    if world == 1 and err:
        if ix is not None:
            ix.free()
        return {"error": err}, None
    local_s = min(runs) if runs else 0.0
    total, gather_s, full = local_s, 0.0, masks
    if world > 1:
        ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=args.coll_device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if ix is not None:
                ix.free()
            return {"error": err or "another rank failed"}, None
        tg = time.perf_counter()
        loc = torch.from_numpy(masks.view(np.int64)).to(args.coll_device)
        full = gather_final_masks(loc, 128)
        torch.cuda.synchronize()
        gather_s = time.perf_counter() - tg
        t = torch.tensor([local_s + gather_s], dtype=torch.float64, device=args.coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        total = float(t.item())
        full = full.cpu().numpy().view(np.uint64)
    device_bytes = int(ix.info.device_bytes)
    ix.free()
    out = {"workload": "BASELINE configs[3]: %d DNA motifs (k = %d) in one call on the 8192-bin x %d-row IBF (h = %d), %d column shard(s)"
                       % (n_motifs, k, rows, h, world),
           "k": k, "motifs": n_motifs, "seconds": total, "queries_per_s": n_motifs / total, "gather_seconds": gather_s,
           "timed_runs_seconds": runs, "failed": int(sum(1 for x in status if x)), "refused_fraction": float(sum(1 for x in status if x)) / n_motifs,
           **stats, "index_build_s": round(build_s, 1), "matrix_bytes_per_gpu": device_bytes, "scaling": "strong (fixed index)",
           **({"collective": {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                              "op": "all_gather of the final masks (%d x %d words per rank)" % (n_motifs, nw)}} if world > 1 else {})}
    if rank == 0:
        ok_q = np.array([s == 0 for s in status])
        found = (full[np.arange(n_motifs), home >> 6] >> (home & 63).astype(np.uint64)) & np.uint64(1)
        if not bool(found[ok_q].all()):
            raise SystemExit("bench: a DNA motif lost the bin it was cut from")
        out["motifs_found_in_their_home_bin"] = int(ok_q.sum())
        out["mean_candidate_bins"] = float(np.unpackbits(full.view(np.uint8), axis=1).sum(axis=1).mean())
        if not args.no_cpu:
            out["cpu_oracle"] = dna_column_check(check_codes, check_word, rows, h, k, motifs, home, status, full, args.cpu_query_seconds)
    return out, {"motifs": motifs, "full": full, "shape": (bins_total, rows, h, k, seq_len)}


def dna_column_check(codes, word, rows, h, k, motifs, home, status, full, budget_s):
    """Word `word` of the gathered masks against the CPU oracle on that 64-bin column: the oracle's own 64-bin IBF of the same
    rows (bin b of it = bin 64*word + b of the index), filled by the oracle's encoder from the sequences' letters.  All motifs
    cut from those bins first, then others until the budget is spent."""
    import oracle as O
    ox = O.Index.ibf(64, rows, h, dna=True, k=k)
    for b in range(64):
        ox.emplace(O.decompose(DNA_LETTERS[codes[b]].tobytes().decode(), k, dna=True), b)
    order = [i for i in range(len(motifs)) if home[i] >> 6 == word] + [i for i in range(len(motifs)) if home[i] >> 6 != word]
    own_total = sum(1 for i in order if home[i] >> 6 == word and not status[i])
    compared, own, t0 = 0, 0, time.perf_counter()
    for i in order:
        if status[i]:
            continue
        want = ox.expected_mask(motifs[i])[0]
        if int(want[0]) != int(full[i, word]):
            raise SystemExit("bench: mask column %d of DNA motif %r differs from the CPU oracle" % (word, motifs[i]))
        compared += 1
        own += int(home[i] >> 6 == word)
        if time.perf_counter() - t0 > budget_s / 2 and own >= own_total:
            break
    dt = time.perf_counter() - t0
    return {"masks_compared": compared, "of_which_cut_from_the_column": own, "column_word": word, "seconds": dt,
            "queries_per_s_on_a_64_bin_column": compared / dt if dt > 0 else None,
            "sample": "mask word %d (64 of the 8192 bins) of the gathered masks; oracle matrix rebuilt on the host from the sequences" % word}


def hibf_descent(capi, torch, dist, args, rank, world, user_bins=65536, children=256, queries=False):
    """Third figure (BASELINE configs[4] shape, SURVEY.md §8d S-HIBF-65536): k-mers/s of the HIBF
    descent — root IBF of 256 merged bins over 256 child IBFs of 256 user bins each, h = 2, sizes from
    compute_bitcount at fpr 0.05, values from a 10-letter k = 5 universe (10^5 k-mers).  Every IBF is
    filled on the device with the real hash and uploaded as a tree; with N ranks the 65536 mask columns
    are sharded (each rank descends only into sub-trees of its own columns).  Runs after the timed
    probe steps; it does not touch `value`."""
    per_bin, h = args.hibf_per_bin, 2
    per_child = user_bins // children
    rng = np.random.default_rng(5)
    shifts = np.uint64(5) * np.arange(4, -1, -1, dtype=np.uint64)

    def values(count):
        return (rng.integers(0, 10, size=(count, 5)).astype(np.uint64) << shifts).sum(axis=1).astype(np.uint64)

    def filled(bins, rows, vals, bins_of):
        ix = capi.Index.create_ibf(bins, rows, h)
        dv = torch.from_numpy(vals.view(np.int64)).cuda()
        db = torch.from_numpy(bins_of.astype(np.uint32).view(np.int32)).cuda()
        ix.emplace_device(dv.data_ptr(), db.data_ptr(), vals.size, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        words = ix.download_words_rows(rows)
        ix.free()
        return words

    with_queries = queries and not args.no_queries
    try:  # (everything up to the query batch is rank-local: a failure here must still let this rank enter the batch's collectives)
        return _hibf_descent_local(capi, torch, dist, args, rank, world, user_bins, children, with_queries, values, filled, per_bin, per_child, h, rng)
    except SystemExit:
        raise
    except Exception as e:  # noqa: BLE001
        out = {"error": repr(e)}
        if with_queries and world > 1:
            out["query_batch"] = hibf_query_batch(torch, dist, args, None, None, user_bins, rank, world, failed=repr(e))
        return out


def _hibf_descent_local(capi, torch, dist, args, rank, world, user_bins, children, with_queries, values, filled, per_bin, per_child, h, rng):
    t0 = time.perf_counter()
    m_child = compute_bitcount(per_bin, 0.05)
    m_root = compute_bitcount(per_bin * per_child, 0.05)
    tb_of = np.repeat(np.arange(per_child, dtype=np.uint32), per_bin)
    descs = [None]
    root_vals, root_bins, sample = [], [], []
    for c in range(children):
        v = values(per_child * per_bin)
        descs.append(dict(bins=per_child, bin_size=m_child, hash_funs=h, words=filled(per_child, m_child, v, tb_of),
                          next_ibf_id=np.zeros(per_child, dtype=np.uint64),
                          tb_to_user=np.arange(c * per_child, (c + 1) * per_child, dtype=np.uint64)))
        root_vals.append(v)
        root_bins.append(np.full(v.size, c, dtype=np.uint32))
        sample.append((v[::997], c * per_child + tb_of[::997]))
    descs[0] = dict(bins=children, bin_size=m_root, hash_funs=h,
                    words=filled(children, m_root, np.concatenate(root_vals), np.concatenate(root_bins)),
                    next_ibf_id=np.arange(1, children + 1, dtype=np.uint64),
                    tb_to_user=np.full(children, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64))
    ix = capi.Index.upload_hibf(user_bins, descs, shard_rank=rank, n_shards=world)
    build_s = time.perf_counter() - t0
    W = ix.shard_words
    n = args.hibf_kmers
    present = np.concatenate([v for v, _ in sample])
    kmers = np.concatenate([np.resize(present, n // 2), values(n - n // 2)])
    rng.shuffle(kmers)
    dk = torch.from_numpy(kmers.view(np.int64)).cuda()
    dm = torch.empty((n, W), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream()
    for _ in range(2):
        ix.probe_device(dk.data_ptr(), n, dm.data_ptr(), None, stream.cuda_stream)
    torch.cuda.synchronize()
    reps = 5
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(reps):
        ix.probe_device(dk.data_ptr(), n, dm.data_ptr(), None, stream.cuda_stream)
    b.record(stream)
    torch.cuda.synchronize()
    dt = a.elapsed_time(b) / 1e3 / reps
    # size-independent check: an inserted value is found in its own user bin (no false negatives)
    pv = np.concatenate([v for v, _ in sample])[:4096]
    pb = np.concatenate([u for _, u in sample])[:4096].astype(np.int64)
    got = ix.probe(pv)
    lo = int(ix.info.shard_word0)
    mine = (pb // 64 >= lo) & (pb // 64 < lo + W)
    bits = (got[np.arange(pv.size)[mine], (pb[mine] // 64 - lo)] >> (pb[mine] % 64).astype(np.uint64)) & np.uint64(1)
    if not bool(bits.all()):
        raise SystemExit("bench: HIBF descent lost an inserted value")
    out = {"workload": "S-HIBF-%d" % user_bins, "user_bins": user_bins, "n_ibf": children + 1, "hash_funs": h, "values_per_bin": per_bin,
           "kmers": n, "kmers_per_s_per_gpu": n / dt, "seconds_per_batch": dt, "mask_bytes_per_kmer": W * 8,
           "mask_write_GBps": n * W * 8 / dt / 1e9, "column_shards": world, "tree_bytes": int(ix.info.device_bytes),
           "index_build_s": round(build_s, 1), "checked_present_values": int(mine.sum())}
    if with_queries:
        out["query_batch"] = hibf_query_batch(torch, dist, args, ix, descs, user_bins, rank, world)
    ix.free()
    return out


def hibf_query_batch(torch, dist, args, ix, descs, user_bins, rank, world, failed=None):
    """BASELINE configs[4]: whole queries on the 65536-user-bin HIBF — Murphy-reduced alphabet, k = 5, 8 KiB masks — the tree
    replicated, its user-bin columns sharded over the ranks (every rank descends only into sub-trees of its own columns), the
    final masks all-gathered.  200 PROSITE-style motifs (the batch of tests/perf_config5_queries.py); a sample of the gathered
    masks is compared with the CPU oracle's HIBF (membership_for restated, oracle/txo_ibf.hpp) built from the same IBFs."""
    from motifs import random_prosite_motifs
    from tetrex_amd.dist import gather_final_masks
    k, reduction = 5, 1
    motifs = random_prosite_motifs(200, 6)
    err, masks, status, stats, runs = failed, None, None, None, []  # (failed: this rank has no index — it only keeps the others company)
    try:
        if err is None:
            ix.query_masks(motifs[:5], False, k, reduction)
    except Exception as e:  # noqa: BLE001
        err = repr(e)
    if world > 1:
        dist.barrier()
    try:
        for _ in range((3 if world == 1 else 1) if err is None else 0):
            t0 = time.perf_counter()
            m_, s_, st_ = ix.query_masks(motifs, False, k, reduction)
            runs.append(time.perf_counter() - t0)
            if masks is None or runs[-1] <= min(runs):
                masks, status, stats = m_, s_, st_
    except Exception as e:  # noqa: BLE001
        err = repr(e)
    if world == 1 and err:
        return {"error": err}
    total, gather_s, full = (min(runs) if runs else 0.0), 0.0, masks
    if world > 1:
        ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=args.coll_device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            return {"error": err or "another rank failed"}
        tg = time.perf_counter()
        full = gather_final_masks(torch.from_numpy(masks.view(np.int64)).to(args.coll_device), int(ix.info.mask_words))
        torch.cuda.synchronize()
        gather_s = time.perf_counter() - tg
        t = torch.tensor([total + gather_s], dtype=torch.float64, device=args.coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        total = float(t.item())
        full = full.cpu().numpy().view(np.uint64)
    out = {"workload": "BASELINE configs[4]: %d PROSITE-style motifs, Murphy alphabet, k = %d, on S-HIBF-%d (%d column shard(s))" % (len(motifs), k, user_bins, world),
           "k": k, "motifs": len(motifs), "seconds": total, "queries_per_s": len(motifs) / total, "gather_seconds": gather_s, "timed_runs_seconds": runs,
           "refused_fraction": float(sum(1 for x in status if x)) / len(motifs), **stats,
           **({"collective": {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                              "op": "all_gather of the final masks (%d x %d words per rank)" % (len(motifs), ix.shard_words)}} if world > 1 else {})}
    if rank == 0:
        out["mean_candidate_bins"] = float(np.unpackbits(full.view(np.uint8), axis=1).sum(axis=1).mean())
        if not args.no_cpu:
            out["cpu_oracle"] = oracle_check_rest(descs, {"kind": "hibf", "bins": user_bins, "dna": False, "k": k, "reduction": reduction}, motifs, full,
                                                  [i for i in range(len(motifs)) if not status[i]], 2 * args.cpu_query_seconds, "S-HIBF-%d batch" % user_bins)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--kmers", type=int, default=1 << 24, help="k-mers per step (batch resident in HBM)")
    ap.add_argument("--bins-per-gpu", type=int, default=1024)
    ap.add_argument("--per-bin", type=int, default=200000, help="values inserted per bin")
    ap.add_argument("--hash", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=1 << 24, help="k-mers timed on the CPU oracle (rank 0, N=1); default: the whole batch")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-query-seconds", type=float, default=10.0, help="CPU oracle time spent on the end-to-end leg's baseline")
    ap.add_argument("--no-queries", action="store_true", help="skip the end-to-end queries/s leg")
    ap.add_argument("--motifs", type=int, default=1000, help="PROSITE-style motifs in the end-to-end batch")
    ap.add_argument("--no-hibf", action="store_true", help="skip the HIBF descent leg")
    ap.add_argument("--no-k6", action="store_true", help="skip the k = 6 end-to-end leg (end_to_end.k6_batch)")
    ap.add_argument("--no-big-batch", action="store_true", help="skip the 10x batch of the end-to-end leg (end_to_end.batch_10x)")
    ap.add_argument("--no-verification", action="store_true", help="skip the end-to-end leg that includes verification (end_to_end.with_verification)")
    ap.add_argument("--hibf-kmers", type=int, default=1 << 20)
    ap.add_argument("--hibf-per-bin", type=int, default=300)
    ap.add_argument("--rows", type=int, default=0, help="override bin_size (rows); >0 selects an out-of-cache variant")
    ap.add_argument("--kmer-bits", type=int, default=20)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: 1024 bins per GPU (index of 1024*N bins), value in shard-probes/s; strong: the fixed 8192-bin x 62.5 M-row "
                         "index cut into N column shards, value in whole-index probes/s")
    ap.add_argument("--no-hbm-leg", action="store_true", help="skip the out-of-cache roofline_hbm leg (8 GB matrix)")
    ap.add_argument("--no-dna-batch", action="store_true", help="skip end_to_end.dna_batch_8192 (BASELINE configs[3]: 10 000 DNA motifs on the 8192-bin index)")
    ap.add_argument("--dna-rows", type=int, default=62500000, help="rows of the 8192-bin DNA index (62.5 M = 64 GB over all shards)")
    ap.add_argument("--dna-motifs", type=int, default=10000)
    ap.add_argument("--dna-seq-len", type=int, default=100000, help="length of the sequence every bin of the DNA index holds")
    ap.add_argument("--rehearse-one-process", action="store_true", help="with --rehearse-single-device: also rehearse the one-process legs (all shards on cuda:0)")
    ap.add_argument("--rehearse-single-device", action="store_true",
                    help="N>1 rehearsal on a one-GPU box: every rank uses cuda:0 and the collectives run over gloo on host "
                         "tensors (RCCL refuses two ranks on one device); numbers from such a run are not bench results")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as a CHILD process (never an exec: nothing
        # here may replace a process), before anything in this one touches a GPU; relay its output and its exit code
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (one rank per GPU: launch with torch.distributed.run --nproc-per-node %d, "
                         "or without a launcher at all)" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.rehearse_single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    args.coll_device = "cpu" if args.rehearse_single_device else "cuda"
    if world > 1:
        # the ranks of a node share its CPUs: each rank's expansion threads = its share (at most 16, at least 2)
        cpus = usable_cpus()
        os.environ.setdefault("TETREX_THREADS", str(max(2, min(64, cpus // world))))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_single_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        import datetime
        host_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(minutes=30))  # host-only waits (rank0_alone)

    from tetrex_amd import capi

    capi.init(local_rank)
    h = args.hash
    strong = args.scaling == "strong"
    if strong:  # S-IBF-8192: fixed index, N column shards
        bins_total = 8192
        if bins_total % (64 * world):
            raise SystemExit("--scaling strong needs a GPU count that divides 128 mask words")
        bins_local = bins_total // world
        if args.rows == 0:
            args.rows = 62500000
        if args.kmer_bits == 20:
            args.kmer_bits = 40
        if args.per_bin == 200000:
            args.per_bin = 20000
        if args.kmers == (1 << 24) and world < 8:
            args.kmers = 1 << 22 if world == 1 else 1 << 23  # the mask batch is kmers x (1024 / N) bytes
    else:
        bins_local = args.bins_per_gpu
        bins_total = bins_local * world
    m = args.rows if args.rows > 0 else compute_bitcount(args.per_bin, 0.05)
    value_bits = args.kmer_bits

    t_build = time.perf_counter()
    ix = build_index(capi, torch, bins_total, bins_local, m, h, rank, world, args.per_bin, value_bits)
    t_build = time.perf_counter() - t_build
    W = ix.shard_words
    assert W == (bins_local + 63) // 64, (W, bins_local)

    n = args.kmers
    kmers_host = splitmix64(2, n) >> np.uint64(64 - value_bits)
    d_kmers = torch.from_numpy(kmers_host.view(np.int64)).cuda()
    d_masks = torch.empty((n, W), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream()
    elapsed, avg_kernel_s = timed_probe_steps(torch, dist, ix, d_kmers, n, d_masks, stream, args.steps, args.warmup, world, args.coll_device)

    bytes_per_probe = h * W * 8 + W * 8 + 8  # rows + mask write + k-mer read (per GPU: its share of the index)
    achieved = bytes_per_probe * n / avg_kernel_s / 1e9
    # strong: a probe is one k-mer against the WHOLE index, answered by all ranks together; weak: one k-mer against one shard
    value = n * args.steps / elapsed if strong else world * n * args.steps / elapsed
    cache_resident = int(ix.info.device_bytes) < (200 << 20)
    traffic, traffic_source = pmc_traffic("S-IBF-1024", n, W, h) if (not strong and args.rows == 0 and args.per_bin == 200000) else (None, None)

    out = {
        "metric": "k-mer IBF probes/sec",
        "value": value,
        "unit": "probes/s" if (strong or world == 1) else "shard-probes/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic" if not args.rehearse_single_device else "synthetic (REHEARSAL: all ranks on one device, gloo collectives; not a bench result)",
        "config": {
            "workload": "S-IBF-8192" if strong else workload_name(bins_local, m, args),
            "bins_per_gpu": bins_local, "bins_total": bins_total, "hash_funs": h, "bin_size_rows": m,
            "kmers_per_step": n, "kmer_bits": value_bits, "values_per_bin": args.per_bin,
            "matrix_bytes_per_gpu": int(ix.info.device_bytes), "mask_words": W,
            "parallelism": "bin-column shards x%d, no collective on the probe path" % world,
            "probe_unit": ("one k-mer against all %d bins (answered by %d shard(s) together)" % (bins_total, world)) if (strong or world == 1)
                          else "one k-mer against one %d-bin shard (a whole-index probe is %d of them)" % (bins_local, world),
            "whole_index_probes_per_s": n * args.steps / elapsed,
            "index_build_s": round(t_build, 2),
        },
        "roofline": {
            "bound": "hbm",
            "resident": ("infinity-cache: the %.0f MB matrix fits the 256 MB MALL, so this is a cache gather rate — see roofline_hbm for the "
                         "out-of-cache figure" % (int(ix.info.device_bytes) / 1e6)) if cache_resident else "hbm (matrix far larger than the 256 MB Infinity Cache)",
            "kernel": "txq::probe_kernel<8,3,2,false,NoRoot>" if (W == 16 and h == 3) else "txq::probe_kernel",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "bytes_per_probe": bytes_per_probe,
            "avg_kernel_ms": avg_kernel_s * 1e3,
            "traffic": traffic,
            **({"traffic_source": traffic_source} if traffic_source else {}),
        },
    }
    if world > 1:
        out["collective"] = {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                             "note": "no collective on the probe path; the end-to-end leg all-gathers the final masks"}

    if rank == 0 and world == 1 and not args.no_cpu and not strong and m < (1 << 24):
        sample = min(args.cpu_sample, n)
        cpu_masks, cpu_rate, cpu_dt = cpu_baseline(ix, m, h, bins_local, kmers_host, sample, threads=1)
        # the timed GPU output doubles as a parity check on the CPU sample
        got = d_masks[:sample].cpu().numpy().view(np.uint64)
        if not np.array_equal(got, cpu_masks):
            raise SystemExit("bench: GPU masks differ from the CPU oracle on the baseline sample")
        out["cpu_baseline"] = {
            "value": cpu_rate, "unit": "probes/s", "cores": 1, "kind": "port",
            "sample": "first %d k-mers of the same batch on the same matrix, single thread (the reference probes single-threaded), %.1f s" % (sample, cpu_dt),
            "host_cpus": os.cpu_count(),
        }
        ncores = min(os.cpu_count() or 1, 16)
        if ncores > 1:
            _, mt_rate, mt_dt = cpu_baseline(ix, m, h, bins_local, kmers_host, sample, threads=ncores)
            out["cpu_baseline_all_cores"] = {"value": mt_rate, "unit": "probes/s", "cores": ncores, "kind": "port",
                                             "sample": "same sample, %d threads, %.1f s" % (ncores, mt_dt)}
        out["parity_checked_probes"] = int(sample)

    if not args.no_queries and not strong:
        args.m_rows = m
        out["end_to_end"] = end_to_end_queries(ix, torch, dist, world, rank, args)
        if world == 1 and not args.no_hibf and "error" not in out["end_to_end"]:
            try:
                out["end_to_end"]["hibf_batch"] = hibf_end_to_end(capi, torch, args)
            except Exception as e:  # noqa: BLE001 - an extra leg must not cost the contract line
                out["end_to_end"]["hibf_batch"] = {"error": repr(e)}
        if world == 1 and not args.no_k6 and "error" not in out["end_to_end"]:
            try:
                out["end_to_end"]["k6_batch"] = k6_end_to_end(capi, torch, args)
            except Exception as e:  # noqa: BLE001 - an extra leg must not cost the contract line
                out["end_to_end"]["k6_batch"] = {"error": repr(e)}
        if world == 1 and not args.no_verification and "error" not in out["end_to_end"]:
            try:
                out["end_to_end"]["with_verification"] = verified_end_to_end(args)
            except Exception as e:  # noqa: BLE001 - an extra leg must not cost the contract line
                out["end_to_end"]["with_verification"] = {"error": repr(e)}

    ix.free()
    if not args.no_hbm_leg and not strong and cache_resident:
        try:
            leg = hbm_leg(capi, torch, args, h)
            if rank == 0:
                out["roofline_hbm"] = leg
        except Exception as e:  # noqa: BLE001 - an extra leg must not cost the contract line
            out["roofline_hbm"] = {"error": repr(e)}
    e2e = out.setdefault("end_to_end", {}) if not args.no_queries else {}
    dna_ctx = None
    if not args.no_queries and not args.no_dna_batch:
        # BASELINE configs[3] in its own shape, at every N and in both scaling modes: 10 000 DNA motifs on the fixed 8192-bin index
        try:
            e2e["dna_batch_8192"], dna_ctx = dna_batch_8192(capi, torch, dist, args, rank, world)
        except SystemExit:
            raise
        except Exception as e:  # noqa: BLE001 - an extra leg must not cost the contract line (a failure BEFORE the leg's collectives
            e2e["dna_batch_8192"] = {"error": repr(e)}  # would leave the other ranks waiting: the leg catches what can fail there itself)
    if not args.no_hibf:
        try:
            # BASELINE configs[4]: descent rate and (queries=True) the motif batch with its gather, at every N and in both modes
            out["hibf"] = hibf_descent(capi, torch, dist, args, rank, world, queries=True)
            if not strong:  # the Swissprot-HIBF shape (BASELINE configs[2]): 1024 user bins, 16 children of 64 bins
                out["hibf_1024"] = hibf_descent(capi, torch, dist, args, rank, world, user_bins=1024, children=16)
        except SystemExit:
            raise
        except Exception as e:  # noqa: BLE001 - an extra leg must not cost the contract line
            out.setdefault("hibf", {"error": repr(e)})
            if not strong:
                out.setdefault("hibf_1024", {"error": repr(e)})
        if world > 1 and not strong:  # the general tree sharded by sub-trees over the ranks (tests/perf_hibf_ragged.py measure_sharded)
            try:
                from perf_hibf_ragged import measure_sharded
                leg = measure_sharded(capi, torch, dist, args.coll_device, rank, world)
                if rank == 0:
                    out["hibf_irregular"] = leg
            except SystemExit:
                raise
            except Exception as e:  # noqa: BLE001
                out["hibf_irregular"] = {"error": repr(e)}
        if world == 1 and rank == 0 and not strong:
            # ... and a GENERAL tree as seqan::hibf's layout shapes the reference's index (include/index_hibf.h:114-129): 65 536 user
            # bins scattered over IBFs of at most 256 technical bins, split bins, user bins next to merged bins (tests/helpers.py
            # layout_hibf).  Plain probes (user-bin order: descent kernels) and a 200-motif batch (session in layout order, against
            # the same batch in user-bin order: masks identical; 2048 probed masks against the CPU oracle).  tests/perf_hibf_ragged.py
            try:
                from perf_hibf_ragged import measure as hibf_irregular
                out["hibf_irregular"] = hibf_irregular(capi, torch, 1 << 20, 256, 65536)
            except SystemExit:
                raise
            except Exception as e:  # noqa: BLE001
                out["hibf_irregular"] = {"error": repr(e)}
    if world > 1 and not args.no_queries and (not args.rehearse_single_device or args.rehearse_one_process):
        # both deployments of the sharded index are measured at first contact with a multi-GPU node: after the one-process-
        # per-GPU legs above, rank 0 alone drives all N devices; everybody else waits on the host (a failure must not cost the line)
        from motifs import random_prosite_motifs
        devs = 1 if args.rehearse_single_device else world  # (rehearsal: txq_init with one device deals every shard onto it)
        kq = max(2, args.kmer_bits // 5)
        if not strong:
            def weak_leg():
                res, _ = one_process_n_devices(capi, torch, devs, world, lambda r: build_index(capi, torch, bins_total, bins_local, m, h, r, world, args.per_bin, value_bits),
                                               random_prosite_motifs(args.motifs, 6), random_prosite_motifs(args.motifs, 8), False, kq,
                                               "the %d-bin index of this run, %d PROSITE-style motifs" % (bins_total, args.motifs))
                return res
            res = rank0_alone(torch, dist, host_group, rank, weak_leg)
            if rank == 0:
                e2e["one_process_n_devices"] = res
        if not args.no_dna_batch:
            def dna_leg():
                bt, rows_, h_, k_, sl = dna_ctx["shape"]
                res, joined = one_process_n_devices(capi, torch, devs, world, lambda r: dna_build_shard(capi, torch, bt, rows_, h_, k_, sl, r, world),
                                                    dna_ctx["motifs"], dna_ctx["motifs"][:200], True, k_,
                                                    "BASELINE configs[3]: %d DNA motifs on the 8192-bin index" % len(dna_ctx["motifs"]))
                if not np.array_equal(joined, dna_ctx["full"]):
                    raise RuntimeError("the host-side join gives other masks than the all-gather of the one-process-per-GPU run")
                res["masks_equal_the_gathered_run"] = True
                return res
            res = rank0_alone(torch, dist, host_group, rank, dna_leg if dna_ctx is not None else (lambda: {"skipped": "the DNA leg failed"}))
            if rank == 0 and isinstance(e2e.get("dna_batch_8192"), dict):
                e2e["dna_batch_8192"]["one_process_n_devices"] = res
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
