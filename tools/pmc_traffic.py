#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, with
--kernel-trace only — MI355X_MICROARCH.md §HBM) of one command.  Usage (on the GPU box):
    tools/pmc_traffic.py <out.json> <tag> -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-queries --no-hibf
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of 16-B/lane coalesced reads
(calibrated here on emplace_kernel, whose input is a known 12 bytes per value) -> corrected = 2 x FETCH + WRITE.
Launches are grouped by kernel name, grid size and BUILD PHASE (a phase starts whenever emplace_kernel launches resume after
other kernels), so the probe launches of the two bench legs — same kernel, same grid, different matrix — stay apart."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys


def one_pass(counter, cmd, outdir):
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", outdir, "-o", "pmc", "--"] + cmd,
                   check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp")
    rows = collections.defaultdict(list)
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            recs = sorted((r for r in csv.DictReader(fh) if r["Counter_Name"] == counter), key=lambda r: int(r["Dispatch_Id"]))
        phase, building = 0, False
        for r in recs:
            name = r["Kernel_Name"].split("(")[0]
            if "emplace_kernel" in name:
                if not building:
                    phase += 1
                building = True
            elif "probe_kernel" in name:
                building = False
            rows[(name, int(r["Grid_Size"]), phase)].append(float(r["Counter_Value"]))
    return rows


def main():
    out_json, tag = sys.argv[1], sys.argv[2]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    cmd = [os.path.abspath(c) if c.endswith(".py") else c for c in cmd]
    fetch = one_pass("FETCH_SIZE", cmd, "/tmp/pmc_fetch_" + tag)
    write = one_pass("WRITE_SIZE", cmd, "/tmp/pmc_write_" + tag)
    kernels = {}
    for key in sorted(set(fetch) | set(write)):
        name, grid, phase = key
        f, w = fetch.get(key, []), write.get(key, [])
        kernels["%s [grid %d, build phase %d]" % (name, grid, phase)] = {
            "FETCH_SIZE_KiB_avg": sum(f) / len(f) if f else None, "FETCH_SIZE_launches": len(f),
            "WRITE_SIZE_KiB_avg": sum(w) / len(w) if w else None, "WRITE_SIZE_launches": len(w)}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) of: " + " ".join(sys.argv[sys.argv.index("--") + 1:]),
               "unit_note": "KiB; gfx950 FETCH_SIZE counts 1/2 of 16-B/lane coalesced reads (MI355X_MICROARCH.md §HBM): corrected traffic = 2 x FETCH + WRITE",
               "kernels": kernels}, open(out_json, "w"), indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
