#!/usr/bin/env python3
"""Sums of the given PMC counters per kernel over one run of a tool (one rocprofv3 --pmc pass, --kernel-trace only).
Usage on the GPU box:  tools/pmc_counters.py out.json tools/k6_profile.py SQ_WAVE_CYCLES SQ_WAIT_ANY ...   (at most 8 SQ counters)"""
import collections, csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, tool, names = sys.argv[1], sys.argv[2], sys.argv[3:]
d = "/tmp/pmc_counters"
env = dict(os.environ, TMPDIR="/tmp", K6_NO_CHECK="1")
subprocess.run(["rocprofv3", "--pmc", *names, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "run", "--", "python3", os.path.join(ROOT, tool)],
               check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp")
tot = collections.defaultdict(collections.Counter)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
res = {k: dict(v) for k, v in tot.items() if any(x in k for x in ("sparse_kernel", "dense_kernel", "probe_kernel"))}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
