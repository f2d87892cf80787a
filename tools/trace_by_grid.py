#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace (csv) per kernel and grid-size bucket: launches, total and mean duration."""
import csv, sys, collections
rows = collections.defaultdict(lambda: [0, 0.0, 0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0][:60]
        wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) if "Grid_Size_X" in r else int(r["Grid_Size"]) // int(r["Workgroup_Size"])
        b = 0
        while (1 << b) < wg:
            b += 1
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        x = rows[(name, b)]
        x[0] += 1; x[1] += d; x[2] += wg
for (name, b), (n, t, wg) in sorted(rows.items()):
    print("%-62s wgs<=2^%-2d launches %5d  total %9.1f us  mean %8.1f us  wgs %9d  us/wg %.3f" % (name, b, n, t, t / n, wg, t / max(1, wg)))
