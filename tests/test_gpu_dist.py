"""Two ranks, one GPU: each rank uploads its bin-column shard of the same index, runs the same
query programs through libtxq, and the final masks are all-gathered (gloo on host tensors here;
RCCL in bench.py on a multi-GPU node).  The reassembled masks must equal the oracle's."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as O
        from tetrex_amd import capi, host
        from tetrex_amd.dist import shard_range, gather_final_masks
        from motifs import PEPTIDE_QUERIES
        capi.init(0)
        bins, m, h, k = 1000, 4099, 3, 4
        ox = O.Index.ibf(bins, m, h, dna=False, k=k)
        rng = np.random.default_rng(5)
        for b in range(bins):
            ox.emplace(rng.integers(0, 1 << 20, size=1500, dtype=np.uint64), b)
        queries = [q_ for q_ in PEPTIDE_QUERIES if "{2,4}" not in q_][:24]
        blob, status, _ = host.compile_batch(queries, False, k, 0, bins)
        ix = capi.Index.upload_ibf(bins, m, h, ox.words(), shard_rank=rank, n_shards=world)
        W = int(ix.info.mask_words)
        assert (int(ix.info.shard_word0), int(ix.info.shard_word0) + ix.shard_words) == shard_range(W, rank, world)
        local = ix.run_programs(blob, len(queries))
        full = gather_final_masks(torch.from_numpy(local.view(np.int64)), W).numpy().view(np.uint64)
        ok = True
        for i, rx in enumerate(queries):
            want, st = ox.query(rx, with_stats=True)
            if st["quirk_merges"] == 0:
                ok = ok and np.array_equal(full[i], want)
        ix.free()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_two_shards_on_one_gpu_gather_to_the_oracle_masks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


SMALL_DNA = ["--dna-rows", "200003", "--dna-motifs", "300", "--dna-seq-len", "3000"]


def _contract_line(res):
    import json
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_rank_rehearsal_prints_one_contract_line():
    """bench.py's N>1 path (shard build per rank, barriers, max-over-ranks timing, final-mask all-gathers of the protein,
    DNA (BASELINE configs[3]) and HIBF (configs[4]) query legs, and the one-process legs) rehearsed with two ranks on the one
    GPU of this box (gloo collectives; RCCL needs one device per rank) — started WITHOUT a launcher: `python bench.py --gpus 2`
    starts its ranks itself as a child process.  Checks the JSON contract, not the numbers."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--kmers", str(1 << 18), "--per-bin", "2000", "--motifs", "40", "--hibf-kmers", str(1 << 16), "--cpu-query-seconds", "2",
           "--rehearse-single-device", "--rehearse-one-process"] + SMALL_DNA
    out = _contract_line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["config"]["bins_total"] == 2048
    # weak mode counts probes of ONE 1024-bin shard; the whole-index rate is spelled out beside it
    assert out["unit"] == "shard-probes/s" and abs(out["value"] - 2 * out["config"]["whole_index_probes_per_s"]) < 1e-6 * out["value"]
    assert out["collective"]["ranks"] == 2 and out["collective"]["backend"] == "gloo"
    e2e = out["end_to_end"]
    assert "error" not in e2e, e2e
    assert e2e["batch"]["failed"] == 0 and e2e["collective"]["ranks"] == 2
    assert e2e["batch"]["gather_seconds"] > 0
    assert "error" not in out["hibf"], out["hibf"]
    assert out["hibf"]["column_shards"] == 2 and out["hibf"]["mask_bytes_per_kmer"] == 4096
    _check_wide_legs(out, 2)
    irr = out["hibf_irregular"]  # the general tree, sharded by sub-trees over the two ranks, masks ORed
    assert "error" not in irr, irr
    assert irr["collective"]["ranks"] == 2 and irr["oracle_masks_compared"] > 20 and irr["queries_layout_order"]["queries_per_s"] > 0
    one = e2e["one_process_n_devices"]
    assert "error" not in one and one["mask_words"] == 32 and one["queries_per_s"] > 0, one


def _check_wide_legs(out, ranks):
    dna = out["end_to_end"]["dna_batch_8192"]  # BASELINE configs[3]: DNA motifs on the fixed 8192-bin index, masks gathered
    assert "error" not in dna and "skipped" not in dna, dna
    assert dna["collective"]["ranks"] == ranks and dna["gather_seconds"] > 0 and dna["failed"] == 0 and dna["k"] == 16
    assert dna["motifs_found_in_their_home_bin"] == dna["motifs"] == 300
    assert dna["cpu_oracle"]["masks_compared"] >= dna["cpu_oracle"]["of_which_cut_from_the_column"] > 0
    one = dna["one_process_n_devices"]
    assert "error" not in one and one["masks_equal_the_gathered_run"] is True and one["mask_words"] == 128, one
    hq = out["hibf"]["query_batch"]  # BASELINE configs[4]: Murphy k = 5 motifs on S-HIBF-65536, masks gathered
    assert "error" not in hq, hq
    assert hq["collective"]["ranks"] == ranks and hq["gather_seconds"] > 0 and hq["k"] == 5 and hq["cpu_oracle"]["masks_compared"] > 0


def test_bench_strong_scaling_rehearsal():
    """--scaling strong: the fixed 8192-bin index cut into N column shards (here N = 2 on one GPU, with fewer rows than
    the real 62.5 M so that the rehearsal builds in seconds): value counts probes of the WHOLE index, bytes_per_probe is
    the per-GPU share; the line carries the configs[3] DNA batch and the configs[4] HIBF batch with their gathers.  Started
    through the external launcher, as the driver does."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--scaling", "strong", "--rows", "2000003", "--per-bin", "500", "--kmers", str(1 << 18), "--hibf-kmers", str(1 << 16),
           "--cpu-query-seconds", "2", "--rehearse-single-device", "--rehearse-one-process"] + SMALL_DNA
    out = _contract_line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT))
    assert out["scaling"] == "strong" and out["unit"] == "probes/s" and out["n_gpus"] == 2
    cfg = out["config"]
    assert cfg["bins_total"] == 8192 and cfg["bins_per_gpu"] == 4096 and cfg["mask_words"] == 64 and cfg["workload"] == "S-IBF-8192"
    assert out["roofline"]["bytes_per_probe"] == 3 * 64 * 8 + 64 * 8 + 8
    assert abs(out["value"] - cfg["whole_index_probes_per_s"]) < 1e-6 * out["value"]
    assert "batch" not in out["end_to_end"] and "hibf_1024" not in out
    _check_wide_legs(out, 2)


def test_bench_refuses_a_rank_count_that_is_not_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode != 0 and "WORLD_SIZE=3" in res.stderr


def test_bench_single_gpu_contract_line_with_all_legs():
    """`python bench.py` at N=1 (small sizes): one JSON line with the contract keys, the roofline and
    cpu_baseline objects, and the two extra legs without errors."""
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--kmers", str(1 << 18), "--per-bin", "2000",
           "--motifs", "40", "--hibf-kmers", str(1 << 16), "--cpu-query-seconds", "1"] + SMALL_DNA
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["metric"] == "k-mer IBF probes/sec" and out["unit"] == "probes/s" and out["n_gpus"] == 1 and out["vs_baseline"] is None
    assert out["higher_is_better"] is True and out["dtype"] == "u64" and out["config"]["workload"].startswith("S-IBF-1024")
    roof = out["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and roof["frac"] > 0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert roof["resident"].startswith("infinity-cache") and ("traffic_source" in roof) == (roof["traffic"] is not None)
    hbm = out["roofline_hbm"]  # the out-of-cache leg: same kernel, 8 GB matrix
    assert "error" not in hbm and hbm["matrix_bytes"] >= 8_000_000_000 and hbm["bytes_per_probe"] == 520 and 0 < hbm["frac"] < 1
    assert hbm["self_check_rows"] > 0 and abs(hbm["frac"] - hbm["achieved"] / hbm["peak"]) < 1e-9
    cpu = out["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0 and out["parity_checked_probes"] == 1 << 18
    e2e = out["end_to_end"]
    assert "error" not in e2e and e2e["batch"]["failed"] == 0 and e2e["cpu_baseline"]["masks_compared"] > 0
    hb = e2e["hibf_batch"]  # BASELINE configs[2]: the batch on a 1024-user-bin HIBF, fused steps checked against the generic descent
    assert "error" not in hb and hb["masks_identical"] is True and hb["failed"] == 0 and hb["queries_per_s"] > 0
    assert "error" not in out["hibf"] and out["hibf"]["checked_present_values"] > 0
    assert "error" not in out["hibf_1024"] and out["hibf_1024"]["user_bins"] == 1024
    dna = e2e["dna_batch_8192"]
    assert "error" not in dna and dna["failed"] == 0 and dna["motifs_found_in_their_home_bin"] == 300 and dna["cpu_oracle"]["masks_compared"] > 0
    assert "error" not in out["hibf"]["query_batch"] and out["hibf"]["query_batch"]["cpu_oracle"]["masks_compared"] > 0
