#!/usr/bin/env python3
"""Timeline of one bench step from a rocprofv3 --kernel-trace CSV: kernels in launch order with
durations and the idle gap before each, then totals (dev tool).  usage: trace_step.py <dir> [step_index]"""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(({"name": r["Kernel_Name"].split("(")[0].replace("rsreg::", ""), "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])}
               for r in csv.DictReader(open(path))), key=lambda r: r["s"])
# a step starts at each k_bbox launch that follows a fused/search kernel (set_target of the next pair)
starts = [i for i, r in enumerate(rows) if r["name"].startswith("k_bbox") and (i == 0 or not rows[i - 1]["name"].startswith("k_bbox"))]
starts = starts[::2] if len(starts) > 1 and any("k_source_keys" in r["name"] for r in rows) else starts
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
a = starts[which]
b = starts[which + 1] if which + 1 < len(starts) else len(rows)
seg = rows[a:b]
t0 = seg[0]["s"]
busy = 0
agg = {}
prev_e = t0
for r in seg:
    d = (r["e"] - r["s"]) / 1e3
    gap = (r["s"] - prev_e) / 1e3
    busy += d
    k = agg.setdefault(r["name"][:60], [0, 0.0, 0.0])
    k[0] += 1
    k[1] += d
    k[2] += max(gap, 0.0)
    prev_e = max(prev_e, r["e"])
span = (seg[-1]["e"] - t0) / 1e3
print("step %d: %d kernels, span %.1f us, kernels busy %.1f us, idle %.1f us" % (which, len(seg), span, busy, span - busy))
for name, (n, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
    print("%-62s x%-3d busy %8.1f us   idle before %8.1f us" % (name, n, d, g))
