#!/usr/bin/env python3
"""Raw timeline of one pair (source load, index build, alignment) from a rocprofv3 --kernel-trace CSV: every kernel with its
start offset, duration, queue and the gap to the kernel before it on the same queue (dev tool).
usage: trace_build.py <dir> [pair_index]"""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(({"name": r["Kernel_Name"].split("(")[0].replace("rsreg::", "").replace("void ", "")[:44], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"]),
                "q": r.get("Queue_Id", "?")} for r in csv.DictReader(open(path))), key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if r["name"].startswith("k_bbox") and not r["name"].startswith("k_bbox_final") and
          (i == 0 or not any(x["name"].startswith("k_bbox") for x in rows[max(0, i - 6):i]))]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
a = starts[which]
b = starts[which + 1] if which + 1 < len(starts) else len(rows)
t0 = rows[a]["s"]
last = {}
for r in rows[a:b]:
    gap = (r["s"] - last[r["q"]]) / 1e3 if r["q"] in last else 0.0
    last[r["q"]] = r["e"]
    print("%8.1f us  +%6.1f  q%-3s gap %6.1f  %s" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, r["q"], gap, r["name"]))
print("span %.1f us" % ((rows[b - 1]["e"] - t0) / 1e3))
