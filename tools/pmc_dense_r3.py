#!/usr/bin/env python3
"""dense_kernel on the end-to-end batch (tools/e2e_profile.py: 1000 PROSITE-style motifs, 1024-bin flat index, h = 3), with rows
gathered (TXQ_KMER_TABLE_MB=0: FlatRows) and through the index's table of all k-mers' masks (TableRows): busy time
(rocprofv3 --kernel-trace --stats), HBM traffic (two --pmc passes, FETCH_SIZE and WRITE_SIZE, separate runs with --kernel-trace
only, as MI355X_MICROARCH.md prescribes; gfx950: traffic = 2 x FETCH + WRITE KiB) and the algorithmic bytes from the session's own
count of the work (TXQ_TRACE): visits x (rows + 1) x mask + suffixes x 2 x mask + zeroed / reduced slots x mask.
Usage on the GPU box:  tools/pmc_dense_r3.py out.json"""
import collections, csv, glob, json, os, re, subprocess, sys

REPS = 4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, outdir, env):
    subprocess.run(["rocprofv3", *args, "--output-format", "csv", "-d", outdir, "-o", "pmc", "--", "python3", os.path.join(ROOT, "tools", "e2e_profile.py")],
                   check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)


def counter(name, outdir, env):
    run(["--pmc", name, "--kernel-trace"], outdir, env)
    tot = collections.Counter()
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "dense_kernel" in r["Kernel_Name"]:
                tot["dense_kernel"] += float(r["Counter_Value"])
    return tot["dense_kernel"]


def main():
    out = {"workload": "tools/e2e_profile.py: %d batches of 1000 PROSITE-style motifs (+ a 10-motif warm-up), 1024-bin index (128-byte masks), h = 3" % REPS, "ways": {}}
    for way, mb, rows in (("rows gathered (FlatRows)", "0", 3), ("table of all k-mers' masks (TableRows)", "512", 1)):
        env = dict(os.environ, TMPDIR="/tmp", REPS=str(REPS), TXQ_KMER_TABLE_MB=mb)
        tag = "t" + mb
        run(["--kernel-trace", "--stats"], "/tmp/pd_stats_" + tag, env)
        busy_ns = launches = 0
        for f in glob.glob("/tmp/pd_stats_%s/**/*kernel_stats.csv" % tag, recursive=True):
            for r in csv.DictReader(open(f)):
                if "dense_kernel" in r["Name"]:
                    busy_ns += int(r["TotalDurationNs"]); launches += int(r["Calls"])
        fetch = counter("FETCH_SIZE", "/tmp/pd_fetch_" + tag, env)
        write = counter("WRITE_SIZE", "/tmp/pd_write_" + tag, env)
        err = subprocess.run(["python3", os.path.join(ROOT, "tools", "e2e_profile.py")], env=dict(env, TXQ_TRACE="1", REPS="1"), capture_output=True, text=True, cwd=ROOT).stderr
        # the full batch's session is the last one (the warm-up session precedes it)
        pairs, suffixes, zeroed, reduced, W = (int(x) for x in re.findall(r"dense work: (\d+) predecessor visits for (\d+) destination suffixes, (\d+) slots zeroed, (\d+) entries reduced; mask (\d+) words", err)[-1])
        mask = W * 8
        algorithmic = pairs * (rows + 1) * mask + suffixes * 2 * mask + zeroed * mask + reduced * mask
        traffic = (2 * fetch + write) * 1024 / REPS
        busy = busy_ns / 1e9 / REPS
        out["ways"][way] = {"per_batch": {"predecessor_visits": pairs, "destination_suffixes": suffixes, "slots_zeroed": zeroed, "entries_reduced": reduced,
                                          "dense_kernel_launches": launches / REPS, "busy_ms": busy * 1e3, "algorithmic_bytes": algorithmic, "hbm_traffic_bytes": traffic,
                                          "traffic_over_algorithmic": traffic / algorithmic},
                            "roofline": {"bound": "hbm", "achieved": algorithmic / busy / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": algorithmic / busy / 8e12, "traffic": traffic}}
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
