#!/usr/bin/env python3
"""Sums the counters of a rocprofv3 --pmc run (csv) per kernel: tools/sum_pmc.py <rocprof output dir> <out.json>.
TCC_HIT_sum / TCC_MISS_sum: the L2 hit rate of a kernel whose rows come out of the caches (MI355X_MICROARCH.md, L2)."""
import collections, csv, glob, json, sys
tot = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
out = {}
for k, c in tot.items():
    h, m = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
    out[k] = dict(c, **({"l2_hit_rate": h / (h + m)} if h + m else {}))
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if "dense_kernel" in k or "probe_kernel" in k}, indent=1))
