#!/usr/bin/env python3
"""Median / min / max duration per (kernel, grid size) from a rocprofv3 --kernel-trace CSV (ksh_* kernels by default).
usage: python tools/trace_rows.py <kernel_trace.csv> [name-substring] [last_n]
With last_n: additionally the mean of the LAST last_n launches of every matching (kernel, grid) pair (bench.py's roofline
replays are the last 96 launches of ksh_env_step_kernel at the 4096-game grid and, since round 3, at the 8192-game grid)."""
import collections
import csv
import sys

want = sys.argv[2] if len(sys.argv) > 2 else "ksh_"
last_n = int(sys.argv[3]) if len(sys.argv) > 3 else 0
d = collections.defaultdict(list)
seq = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if want in n:
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        d[(n[:64], int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1))].append(us)
        seq[(n[:64], int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1))].append((int(r["Start_Timestamp"]), us))
print("%-66s %9s %5s %9s %9s %9s" % ("kernel", "grid", "n", "median_us", "min_us", "max_us"))
for k, v in sorted(d.items()):
    v = sorted(v)
    print("%-66s %9d %5d %9.1f %9.1f %9.1f" % (k[0], k[1], len(v), v[len(v) // 2], v[0], v[-1]))
if last_n:
    print()
    for (n, grid), v in sorted(seq.items()):
        v = [us for _, us in sorted(v)][-last_n:]
        print("%-66s grid %9d last %4d launches: mean %.2f us" % (n, grid, len(v), sum(v) / len(v)))
