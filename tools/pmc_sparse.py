#!/usr/bin/env python3
"""HBM traffic and busy time of sparse_kernel (the pushed steps of tracked blocks) on the k = 6 end-to-end batch
(tools/k6_profile.py, K6_NO_CHECK=1: index build, one warm batch, three timed batches of 200 motifs).  Three rocprofv3 runs:
--kernel-trace --stats for the durations, and two --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only,
as MI355X_MICROARCH.md prescribes; gfx950: traffic = 2 x FETCH + WRITE KiB, the correction calibrated in tools/pmc_traffic.py).
The kernel gathers 16 bytes per lane from random 128-byte rows: its roofline is the HBM's.
Usage on the GPU box:  tools/pmc_sparse.py out.json"""
import collections, csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BATCHES = 4  # warm + three timed


def run(args, outdir):
    env = dict(os.environ, TMPDIR="/tmp", K6_NO_CHECK="1")
    subprocess.run(["rocprofv3", *args, "--output-format", "csv", "-d", outdir, "-o", "k6", "--", "python3", os.path.join(ROOT, "tools", "k6_profile.py")],
                   check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp")


def counter(name, outdir):
    run(["--pmc", name, "--kernel-trace"], outdir)
    tot = collections.Counter()
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                tot[r["Kernel_Name"].split("(")[0].split("<")[0]] += float(r["Counter_Value"])
    return tot


def algorithmic_bytes():
    """TXQ_TRACE: every session reports what its pushed steps amounted to (sparse_units_kernel's device-side counters: live
    entries, entry x residue units, non-empty products, units that left a bit) and the bytes that is — per entry its list
    index and mask, per unit h row segments, per product a 16-byte read-modify-write, per unit with a bit a bitmap word."""
    import re
    env = dict(os.environ, TMPDIR="/tmp", K6_NO_CHECK="1", TXQ_TRACE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "k6_profile.py")], env=env, capture_output=True, text=True, cwd="/tmp")
    total, detail = 0.0, []
    for m in re.finditer(r"sparse steps: (\d+) live entries, (\d+) units \(entry x residue\), (\d+) non-empty \d+-byte products, (\d+) units that left a bit; algorithmic bytes (\d+)", r.stderr):
        total += float(m.group(5))
        detail.append({"live_entries": int(m.group(1)), "units": int(m.group(2)), "products": int(m.group(3)), "units_with_a_bit": int(m.group(4)), "bytes": int(m.group(5))})
    return total, detail


def main():
    run(["--kernel-trace", "--stats"], "/tmp/pmc_sparse_stats")
    stats = {}
    for f in glob.glob("/tmp/pmc_sparse_stats/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            stats[r["Name"].split("(")[0].split("<")[0]] = (int(r["Calls"]), int(r["TotalDurationNs"]))
    fetch = counter("FETCH_SIZE", "/tmp/pmc_sparse_fetch")
    write = counter("WRITE_SIZE", "/tmp/pmc_sparse_write")
    out = {"workload": "tools/k6_profile.py (K6_NO_CHECK=1): %d batches of 200 PROSITE-style motifs at k = 6 on the 1024-bin Swissprot-shaped flat IBF" % BATCHES, "kernels": {}}
    for key in stats:
        if not any(x in key for x in ("sparse_units_kernel", "sparse_kernel", "sparse_plan", "clear_blocks", "exec_units", "probe_kernel")):
            continue
        calls, ns = stats[key]
        traffic = (2 * fetch.get(key, 0) + write.get(key, 0)) * 1024
        out["kernels"][key] = {"launches_per_batch": calls / BATCHES, "busy_ms_per_batch": ns / 1e6 / BATCHES, "hbm_traffic_bytes_per_batch": traffic / BATCHES,
                               "achieved_GBps": traffic / (ns / 1e9) / 1e9 if ns else None}
    k = [x for x in out["kernels"] if "sparse_units_kernel" in x]
    if k:
        a = out["kernels"][k[0]]
        alg, detail = algorithmic_bytes()
        busy_s = a["busy_ms_per_batch"] * BATCHES / 1e3
        traffic = a["hbm_traffic_bytes_per_batch"] * BATCHES
        out["algorithmic"] = {"bytes_all_batches": alg, "sessions": detail,
                              "what": "device-side counters under TXQ_TRACE (a separate, unprofiled run of the same command): per live entry 4 B of list + its mask, per "
                                      "(entry, residue) unit h row segments of the mask's width, per non-empty 16-byte product a read-modify-write, per unit that left a bit one bitmap word"}
        out["roofline"] = {"kernel": "txq::sparse_units_kernel<3,true,3,FlatRows>", "bound": "hbm", "achieved": alg / busy_s / 1e9 if busy_s else None, "peak": 8000.0, "unit": "GB/s",
                           "frac": alg / busy_s / 1e9 / 8000.0 if busy_s else None, "traffic": traffic / BATCHES, "algorithmic_bytes_per_batch": alg / BATCHES,
                           "traffic_over_algorithmic": traffic / alg if alg else None,
                           "note": "achieved = algorithmic bytes / busy time of the kernel (rocprofv3 --kernel-trace --stats); traffic = 2 x FETCH_SIZE + WRITE_SIZE of "
                                   "separate --pmc passes (the matrix is 160 MB: Infinity-Cache hits are counted by these counters)"}
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
