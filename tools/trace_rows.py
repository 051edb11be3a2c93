#!/usr/bin/env python3
"""Median / min / max duration per (kernel, grid size) from a rocprofv3 --kernel-trace CSV (ksh_* kernels by default).
usage: python tools/trace_rows.py <kernel_trace.csv> [name-substring]"""
import collections
import csv
import sys

want = sys.argv[2] if len(sys.argv) > 2 else "ksh_"
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if want in n:
        d[(n[:64], int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-66s %9s %5s %9s %9s %9s" % ("kernel", "grid", "n", "median_us", "min_us", "max_us"))
for k, v in sorted(d.items()):
    v = sorted(v)
    print("%-66s %9d %5d %9.1f %9.1f %9.1f" % (k[0], k[1], len(v), v[len(v) // 2], v[0], v[-1]))
