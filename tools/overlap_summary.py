#!/usr/bin/env python3
"""Dev probe: reads a rocprofv3 --kernel-trace CSV of `bench.py --workload chain --in-flight K` and says how the search launches of
the alignments in flight lie in time: per stream the k_icp_fused_dense launches, and over the run the share of the time during which
0 / 1 / 2 / ... search launches (and 0 / 1 / ... kernels of any kind) were executing.
  python tools/overlap_summary.py <kernel_trace.csv> [<skip seconds of warm-up>]"""
import csv
import sys
from collections import defaultdict

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
ev = []
per_stream = defaultdict(list)
for r in rows:
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3     # us
    search = "k_icp_fused_dense" in r["Kernel_Name"]
    ev.append((a, b, search, r["Stream_Id"]))
    if search:
        per_stream[r["Stream_Id"]].append((a, b))
# the steady part: from the middle of the run on (warm-up, allocations and the sequential cross-check lie outside)
lo = sorted(a for a, _, s, _ in ev if s)[len([1 for e in ev if e[2]]) // 3]
hi = sorted(b for _, b, s, _ in ev if s)[2 * len([1 for e in ev if e[2]]) // 3]
print("kernel dispatches: %d, of them search launches: %d on %d streams; window looked at: %.1f .. %.1f ms of the run" %
      (len(ev), sum(1 for e in ev if e[2]), len(per_stream), lo / 1e3, hi / 1e3))
for sid, l in sorted(per_stream.items()):
    w = [(a, b) for a, b in l if a >= lo and b <= hi]
    if w:
        print("  stream %s: %d search launches in the window, mean %.1f us, busy %.0f %% of the window" %
              (sid, len(w), sum(b - a for a, b in w) / len(w), 100 * sum(b - a for a, b in w) / (hi - lo)))
for what, pick in (("search launches", lambda e: e[2]), ("kernels of any kind", lambda e: True)):
    pts = []
    for a, b, s, sid in ev:
        if pick((a, b, s, sid)) and b > lo and a < hi:
            pts.append((max(a, lo), 1))
            pts.append((min(b, hi), -1))
    pts.sort()
    depth, last, share = 0, lo, defaultdict(float)
    for t, d in pts:
        share[depth] += t - last
        last = t
        depth += d
    share[depth] += hi - last
    print("%s executing at once, share of the window: " % what + ", ".join("%d: %.0f %%" % (k, 100 * v / (hi - lo)) for k, v in sorted(share.items())))
# one stretch of the timeline, launch by launch
print("a stretch of the window (us since its start; stream: start - end of every search launch):")
shown = sorted((a, b, sid) for a, b, s, sid in ev if s and a >= lo)[:24]
for a, b, sid in shown:
    print("  stream %s: %8.1f - %8.1f  %s" % (sid, a - lo, b - lo, " " * int((a - lo) / 8) + "#" * max(1, int((b - a) / 8))))
