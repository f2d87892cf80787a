#!/usr/bin/env python3
"""HBM traffic of the dense step kernel on the end-to-end batch, against its algorithmic bytes.  Two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE; separate runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes; gfx950: FETCH_SIZE
counts half of the bytes of 16-B/lane reads -> traffic = 2 x FETCH + WRITE, the correction calibrated in
tools/pmc_traffic.py) of `tools/e2e_profile.py` (REPS batches of 1000 motifs), summed over all dense_kernel launches
and divided by the batches; the algorithmic side from the session's own count of the work (TXQ_TRACE).
Usage on the GPU box:  tools/pmc_dense.py out.json"""
import collections, csv, glob, json, os, re, subprocess, sys

REPS = 4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_pass(counter, outdir):
    env = dict(os.environ, TMPDIR="/tmp", REPS=str(REPS))
    subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", outdir, "-o", "pmc", "--", "python3",
                    os.path.join(ROOT, "tools", "e2e_profile.py")], check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
    tot = collections.Counter()
    n = collections.Counter()
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                name = r["Kernel_Name"].split("(")[0].split("<")[0]
                tot[name] += float(r["Counter_Value"])
                n[name] += 1
    return tot, n


def main():
    fetch, nf = one_pass("FETCH_SIZE", "/tmp/pmc_dense_fetch")
    write, nw = one_pass("WRITE_SIZE", "/tmp/pmc_dense_write")
    env = dict(os.environ, TXQ_TRACE="1", REPS="1")
    err = subprocess.run(["python3", os.path.join(ROOT, "tools", "e2e_profile.py")], env=env, capture_output=True, text=True, cwd=ROOT).stderr
    m = [x for x in re.findall(r"dense work: (\d+) predecessor visits for (\d+) destination suffixes, (\d+) slots zeroed, (\d+) entries reduced; mask (\d+) words", err)]
    pairs, suffixes, zeroed, reduced, W = (int(x) for x in m[-1])  # the full batch's session (the warm-up session precedes it)
    h = 3
    mask = W * 8
    algorithmic = pairs * (h + 1) * mask + suffixes * 2 * mask + zeroed * mask + reduced * mask
    key = [k for k in fetch if "dense_kernel" in k][0]
    traffic = (2 * fetch[key] + write[key]) * 1024 / REPS
    out = {"kernel": "txq::dense_kernel<3,true,3,FlatRows>", "workload": "tools/e2e_profile.py: bench batch of 1000 PROSITE-style motifs, 1024-bin index (128-byte masks), h = 3",
           "per_batch": {"predecessor_visits": pairs, "destination_suffixes": suffixes, "slots_zeroed": zeroed, "entries_reduced": reduced,
                         "algorithmic_bytes": algorithmic,
                         "algorithmic_note": "visits x (h + 1) x 128 B (h row segments + the predecessor's mask) + suffixes x 2 x 128 B (destination read-modify-write) + zeroed / reduced slots x 128 B",
                         "dense_kernel_launches": nf[key] / REPS, "FETCH_SIZE_KiB": fetch[key] / REPS, "WRITE_SIZE_KiB": write[key] / REPS,
                         "hbm_traffic_bytes": traffic, "traffic_over_algorithmic": traffic / algorithmic},
           "unit_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (with --kernel-trace only); gfx950 FETCH_SIZE counts 1/2 of 16-B/lane reads: traffic = 2 x FETCH + WRITE; "
                        "divided by the %d full batches of the run (its 10-motif warm-up batch is included in the sums: < 1 %%)" % REPS}
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    print(json.dumps(out["per_batch"], indent=1))


if __name__ == "__main__":
    main()
