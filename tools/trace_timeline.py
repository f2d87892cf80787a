#!/usr/bin/env python3
"""Busy / idle breakdown of the GPU during the LAST batch of tools/e2e_profile.py from a rocprofv3 kernel trace (csv):
launches, busy time per kernel, and the gaps between consecutive kernels (launch-bound time)."""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:50],
                     int(r.get("Grid_Size_X", 0) or 0) // max(int(r.get("Workgroup_Size_X", 1) or 1), 1), r.get("Queue_Id", "")))
rows.sort()
# The tools run their batches back to back: between two batches the device idles while the host expands the next batch's first
# wave (>= 0.9 ms); inside a batch the gaps are shorter.  The last stretch of at least 30 launches without such a gap = the last batch.
cuts = [0] + [i for i in range(1, len(rows)) if rows[i][0] - max(r[1] for r in rows[max(0, i - 8):i]) > 900_000] + [len(rows)]
seg = rows[cuts[-2]:cuts[-1]]
for a, b in zip(reversed(cuts[:-1]), reversed(cuts[1:])):
    if b - a >= 30:
        seg = rows[a:b]
        break
span = (seg[-1][1] - seg[0][0]) / 1e3
busy = collections.Counter(); n = collections.Counter()
gap_hist = collections.Counter(); gaps = 0.0
for i, (s, e, name, *_) in enumerate(seg):
    busy[name] += (e - s) / 1e3; n[name] += 1
    if i:
        g = (s - seg[i - 1][1]) / 1e3
        if g > 0:
            gaps += g
            gap_hist[min(int(g // 5) * 5, 100)] += 1
# (two streams: kernels of consecutive waves overlap — the device is busy for the UNION of their intervals)
union, cur_s, cur_e = 0.0, seg[0][0], seg[0][1]
for s_, e_, *_ in seg[1:]:
    if s_ > cur_e:
        union += (cur_e - cur_s) / 1e3
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
union += (cur_e - cur_s) / 1e3
print("last batch: %d launches over %.1f us; device busy %.1f us (kernel durations add up to %.1f us: waves overlap), idle %.1f us" % (len(seg), span, union, sum(busy.values()), span - union))
for name, t in busy.most_common():
    print("  %-52s %5d launches %9.1f us" % (name, n[name], t))
print("  gaps by size (us):", sorted(gap_hist.items()))
# where the long gaps are: time since the batch's first launch, what ended before and what began after
t0 = seg[0][0]
for i in range(1, len(seg)):
    g = (seg[i][0] - max(r[1] for r in seg[:i])) / 1e3
    if g >= 100:
        print("  idle %.0f us at +%.0f us: after %s, before %s" % (g, (seg[i][0] - t0) / 1e3, seg[i - 1][2][:40], seg[i][2][:40]))

# the dense / sparse launches one by one, when asked (argv[2] = "launches"): start, duration, workgroups, queue
if len(sys.argv) > 2 and sys.argv[2] == "launches":
    for s, e, name, wgs, q in seg:
        if "dense_kernel" in name or "sparse_kernel" in name:
            print("  +%8.1f us  %7.1f us  %7d workgroups  queue %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, wgs, q, name[10:36]))
