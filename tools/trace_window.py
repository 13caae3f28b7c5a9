#!/usr/bin/env python3
"""usage: tools/trace_window.py <rocprofv3 kernel_trace.csv>  -- splits the trace into calls at k_seg_count, keeps the calls whose k_lift_lanes
grid is the most common small one (the window_50k calls), prints the mean timeline of a call: kernel, start, duration, gap to the one before"""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort()
calls, cur = [], []
for r in rows:
    if r[2] == "k_seg_count" and cur:
        calls.append(cur)
        cur = []
    cur.append(r)
if cur:
    calls.append(cur)
sig = collections.Counter(tuple(k[2] for k in c) for c in calls[-30:])
want = sig.most_common(1)[0][0]
sel = [c for c in calls[-30:] if tuple(k[2] for k in c) == want]
print(f"{len(calls)} calls in the trace; {len(sel)} of the last 30 have the most common kernel sequence ({len(want)} kernels)")
n = len(sel)
print(f"{'kernel':28s} {'start us':>9s} {'dur us':>8s} {'gap us':>8s}")
tot_d = tot_g = 0.0
for i, name in enumerate(want):
    st = sum(c[i][0] - c[0][0] for c in sel) / n / 1e3
    du = sum(c[i][1] - c[i][0] for c in sel) / n / 1e3
    gp = sum((c[i][0] - c[i - 1][1]) if i else 0 for c in sel) / n / 1e3
    tot_d += du
    tot_g += gp
    print(f"{name:28s} {st:9.1f} {du:8.1f} {gp:8.1f}")
span = sum(c[-1][1] - c[0][0] for c in sel) / n / 1e3
per = sum((sel[j + 1][0][0] - sel[j][0][0]) for j in range(n - 1)) / max(1, n - 1) / 1e3
print(f"span of a call {span:.1f} us = kernels {tot_d:.1f} + gaps {tot_g:.1f}; call to call {per:.1f} us")
