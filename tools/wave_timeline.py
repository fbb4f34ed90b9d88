#!/usr/bin/env python3
"""Dev probe: how many waves of the LAST fused search launch are resident over time (GPU only).

Runs one bench pair with RSREG_WAVE_TIMES + RSREG_WAVE_TIMES_LIGHT (start/end stamp and hardware slot of every
wave at the product kernel's own occupancy) and prints the residency timeline, per-XCD spans and the
distribution of wave durations.  wall_clock64 ticks at 100 MHz."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
path = os.path.join(tempfile.gettempdir(), "rsreg_wave_light.bin")
os.environ["RSREG_WAVE_TIMES"] = path
os.environ["RSREG_WAVE_TIMES_LIGHT"] = "1"
os.environ["RSREG_DIAG"] = "1"   # (per-launch times, wave stamps and dumps are the diagnostic build's: librsreg_diag.so, csrc/tunables.hpp)
import rsreg_amd  # noqa: E402,F401
from rsreg_amd import api, synth  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "N1M"
if "x" in size:
    size = tuple(int(v) for v in size.split("x"))   # WxH
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tgt, src = synth.render_frame(0, size, "bench"), synth.render_frame(1, size, "bench")
guess = synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
icp = api.IterativeClosestPoint(api.Context(0, profiling=True))
icp.params = api.icp_params(max_iterations=iters, criteria_mode=1, pipeline_mode=2, max_correspondence_distance=0.05)
icp.setInputSource(src)
icp.setInputTarget(tgt)
icp.align(guess)
icp.align(guess)
raw = np.fromfile(path, dtype=np.uint64)
nw = len(raw) // 16
raw = raw[: 16 * nw].reshape(nw, 16)
ok = (raw[:, 10] >> np.uint64(63)) == 1
raw = raw[ok]
t0, t1 = raw[:, 0].astype(np.int64), raw[:, 4].astype(np.int64)
base = t0.min()
us0, us1 = (t0 - base) / 100.0, (t1 - base) / 100.0
hw = raw[:, 10] & np.uint64(0xffffffff)
xcc = (raw[:, 11] & np.uint64(0xf)).astype(int)
simd, cu, sh, se = (hw >> np.uint64(4)) & np.uint64(3), (hw >> np.uint64(8)) & np.uint64(15), (hw >> np.uint64(12)) & np.uint64(1), (hw >> np.uint64(13)) & np.uint64(7)
print("waves %d (of %d), span %.1f us, mean wave %.1f us, p50 %.1f p90 %.1f p99 %.1f max %.1f" %
      (len(raw), nw, us1.max(), (us1 - us0).mean(), *np.percentile(us1 - us0, [50, 90, 99]), (us1 - us0).max()))
print("sum of wave time / span = %.0f resident waves on average" % ((us1 - us0).sum() / us1.max()))
lgc = ((raw[:, 11] >> np.uint64(8)) & np.uint64(15)).astype(int)
for c in sorted(set(lgc.tolist())):
    m = lgc == c
    d = (us1 - us0)[m]
    print("  waves of tiles searched by %d lane(s) per query: %6d, duration mean %.1f p50 %.1f p90 %.1f max %.1f us, started p50 %.1f max %.1f us" %
          (1 << c, m.sum(), d.mean(), np.percentile(d, 50), np.percentile(d, 90), d.max(), np.percentile(us0[m], 50), us0[m].max()))
# what bench.py quotes (roofline_issue.tail_fraction, .vmem_frac_while_full): RSREG_WAVE_TIMELINE_JSON=<file>
if os.environ.get("RSREG_WAVE_TIMELINE_JSON"):
    import json
    grid_t = np.linspace(0.0, us1.max(), 400)
    resident = np.array([((us0 <= t) & (us1 > t)).sum() for t in grid_t])
    full = grid_t[resident >= 0.9 * 8192]
    json.dump({"size": str(size), "iterations": iters, "waves": int(len(raw)), "slots": 8192, "span_us": float(us1.max()),
               "sum_wave_time_us": float((us1 - us0).sum()), "mean_wave_us": float((us1 - us0).mean()),
               "full_until_us": float(full.max()) if len(full) else 0.0,
               "source": "python tools/wave_timeline.py %s %d (RSREG_WAVE_TIMES_LIGHT: start / end stamp of every wave of the last launch)" % (size, iters)},
              open(os.environ["RSREG_WAVE_TIMELINE_JSON"], "w"), indent=1)
slots = len(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist(), simd.tolist())))
print("distinct (xcc, se, sh, cu, simd): %d; distinct (xcc,se,sh,cu): %d" %
      (slots, len(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())))))
step = max(us1.max() / 24, 1.0)
print("  t(us)  resident  started  finished")
for k in range(int(np.ceil(us1.max() / step))):
    a, b = k * step, (k + 1) * step
    mid = (a + b) / 2
    print("%7.1f %9d %8d %9d" % (a, int(((us0 <= mid) & (us1 > mid)).sum()), int(((us0 >= a) & (us0 < b)).sum()),
                                 int(((us1 >= a) & (us1 < b)).sum())))
print("per XCD: waves, first start, last end (us)")
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print("  xcc %d: %6d  %7.1f  %7.1f   sum wave time %.0f us" % (x, m.sum(), us0[m].min(), us1[m].max(), (us1[m] - us0[m]).sum()))
# start time by wave index: does the dispatcher keep up?
idx = np.nonzero(ok)[0]
for q in (0.1, 0.25, 0.5, 0.75, 0.9, 1.0):
    k = min(int(q * len(idx)), len(idx) - 1)
    print("  wave #%6d started at %7.1f us" % (idx[k], us0[k]))

# ---- what would splitting the heaviest tiles buy?  List-scheduling model on the measured wave durations
# (a wave keeps its measured duration wherever it runs: contention effects are ignored).
import heapq


def simulate(dur, slots=8192):
    free = [0.0] * slots
    heapq.heapify(free)
    end = 0.0
    for d in dur:
        t = heapq.heappop(free)
        heapq.heappush(free, t + d)
        end = max(end, t + d)
    return end


dur_all = np.zeros(nw)
dur_all[ok] = us1 - us0
dur = dur_all[ok]
print("model: as launched %.1f us (measured span %.1f)" % (simulate(dur), us1.max()))
print("model: longest first %.1f us" % simulate(np.sort(dur)[::-1]))
tile = dur_all[: (nw // 2) * 2].reshape(-1, 2).max(axis=1)
for frac in (0.01, 0.03, 0.1, 0.2):
    for L in (2, 4, 8):
        thr = np.quantile(tile, 1.0 - frac)
        heavy = tile >= thr
        over = 3.0   # us of fixed cost per sub-block (load, transform, seed, merge, last-block sums)
        parts = np.repeat(tile[heavy] / L + over, 2 * L)
        rest = dur_all[: (nw // 2) * 2].reshape(-1, 2)[~heavy].ravel()
        # `rest` as tiles in 4 cost classes (longest class first, natural order inside a class)
        rt = dur_all[: (nw // 2) * 2].reshape(-1, 2)[~heavy]
        cls = np.digitize(rt.max(axis=1), np.quantile(rt.max(axis=1), [0.25, 0.5, 0.75]))
        classed = np.concatenate([rt[cls == c].ravel() for c in (3, 2, 1, 0)])
        print("model: heaviest %4.0f %% of tiles (>= %.0f us) over %d lanes per query: heavy first %.1f us, all longest-first %.1f us, 4 cost classes %.1f us; total wave time x%.2f" %
              (100 * frac, thr, L, simulate(np.concatenate([np.sort(parts)[::-1], rest])),
               simulate(np.sort(np.concatenate([parts, rest]))[::-1]), simulate(np.concatenate([np.sort(parts)[::-1], classed])),
               (parts.sum() + rest.sum()) / dur.sum()))
