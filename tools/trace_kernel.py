#!/usr/bin/env python3
"""Every launch of the kernels whose name contains <substr> in a rocprofv3 --kernel-trace CSV: duration and grid size (dev tool).
usage: trace_kernel.py <dir> <substr> [max_rows]"""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
sub = sys.argv[2]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 80
rows = [r for r in csv.DictReader(open(path)) if sub in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows[:lim]:
    g = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
    print("%8.1f us  grid %s  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, g, r["Kernel_Name"].split("(")[0][-40:]))
print(len(rows), "launches")
