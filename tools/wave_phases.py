#!/usr/bin/env python3
"""Dev probe: where the waves of the fused search kernel spend their time (GPU only).
Runs one pair with RSREG_WAVE_TIMES set (the diagnostic instantiation of k_icp_fused_dense
stamps every wave) and prints the per-phase distribution of the last launch."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
path = os.path.join(tempfile.gettempdir(), "rsreg_wave_times.bin")
os.environ["RSREG_WAVE_TIMES"] = path
os.environ["RSREG_DIAG"] = "1"   # (per-launch times, wave stamps and dumps are the diagnostic build's: librsreg_diag.so, csrc/tunables.hpp)
import rsreg_amd  # noqa: E402
from rsreg_amd import api, synth  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "N1M"
preset = sys.argv[2] if len(sys.argv) > 2 else "bench"
gate = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
tgt, src = synth.render_frame(0, size, preset), synth.render_frame(1, size, preset)
guess = synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
icp = api.IterativeClosestPoint(api.Context(0, profiling=True))
icp.params = api.icp_params(max_iterations=iters, criteria_mode=1, max_correspondence_distance=gate)
icp.setInputSource(src)
icp.setInputTarget(tgt)
icp.align(guess)
raw = np.fromfile(path, dtype=np.uint64)
gi = icp.grid_info()
nq = int(gi.n_source_distinct)
nw = (nq + 63) // 64
w = raw[:16 * nw].reshape(-1, 16).astype(np.int64)
lane = raw[16 * nw:].view(np.uint32)[:nq]
own, r1, frow, fscan = (lane & 255).astype(np.int64), (lane >> 8 & 255).astype(np.int64), (lane >> 16 & 255).astype(np.int64), (lane >> 24).astype(np.int64)
near = own + r1
far = 2 * frow + fscan            # a far row costs about two dependent loads besides its scans


def wave_steps(v, order=None):
    """Sum over waves of the slowest lane's steps (what a wave executes), for lanes taken in `order`."""
    if order is not None:
        v = v[order]
    pad = (-len(v)) % 64
    v = np.concatenate([v, np.zeros(pad, v.dtype)]).reshape(-1, 64)
    return v.max(axis=1)


cost = near + far
print("lanes %d: steps per lane mean near %.2f far %.2f; lanes needing far rows %.2f %%" % (nq, near.mean(), far.mean(), 100.0 * (frow > 0).mean()))
print("per-lane cost percentiles p50/p90/p99/p99.9/max: %s" % [int(np.percentile(cost, p)) for p in (50, 90, 99, 99.9, 100)])
base = wave_steps(near).sum() + wave_steps(far).sum()
print("wave-steps as launched: near %d + far %d (sum of lanes / 64 would be %d)" % (wave_steps(near).sum(), wave_steps(far).sum(), cost.sum() // 64))
print("one flat loop over all phases (a wave pays the maximum of its lanes' totals): own %d + ring-1 scans %d + far %d as three loops -> %d as one (%.2f x)"
      % (wave_steps(own).sum(), wave_steps(r1).sum(), wave_steps(far).sum(), wave_steps(cost).sum(),
         (wave_steps(own).sum() + wave_steps(r1).sum() + wave_steps(far).sum()) / max(wave_steps(cost).sum(), 1)))
for win in (128, 512, 1024, 4096, 1 << 30):
    key = (np.arange(nq) // win) * 100000 - np.minimum(cost, 99999)
    o = np.argsort(key, kind="stable")
    ws = wave_steps(cost, o)
    print("lanes regrouped by cost inside windows of %d queries: wave-steps %d (%.2f x), slowest wave %d steps"
          % (win, ws.sum(), base / max(ws.sum(), 1), ws.max()))
for cap in (8, 12, 16, 24, 32):
    k = np.maximum(1, np.ceil(cost / cap)).astype(np.int64)
    k = np.minimum(1 << np.ceil(np.log2(k)).astype(np.int64), 64)
    per = np.ceil(cost / k)
    lanes = int(k.sum())
    key = (np.arange(nq) // 1024) * 100000 - np.minimum(cost, 99999)
    o = np.argsort(key, kind="stable")
    v = np.repeat(per[o], k[o])
    print("  + queries above %d steps split over 2..64 lanes: %d lanes (%.3f x), wave-steps %d (%.2f x), slowest wave %d steps"
          % (cap, lanes, lanes / nq, wave_steps(v).sum(), base / wave_steps(v).sum(), wave_steps(v).max()))
t0 = w[:, 0].min()
span = (w[:, 4].max() - t0) / 100.0
print("waves %d, kernel span %.1f us (launch avg by events %.1f us)" % (len(w), span, icp.result.ms_nn / icp.result.n_nn_launches * 1e3))
ph = {"rings 0-1 (+load, seed)": w[:, 1] - w[:, 0], "far rings": w[:, 2] - w[:, 1], "store/wait": w[:, 3] - w[:, 2],
      "terms + block reduce": w[:, 4] - w[:, 3], "whole wave": w[:, 4] - w[:, 0]}
for k, v in ph.items():
    v = v / 100.0
    print("%-26s mean %7.2f us  p50 %7.2f  p90 %7.2f  p99 %7.2f  max %7.2f   share of wave time %.1f %%"
          % (k, v.mean(), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max(),
             100.0 * v.sum() / (w[:, 4] - w[:, 0]).sum() * 100.0 / 100.0))
names = ["own-cell steps", "ring-1 cells", "ring-1 scan steps", "far rows", "far scan steps"]
dur = (w[:, 4] - w[:, 0]) / 100.0
heavy = dur >= np.percentile(dur, 99)
for k, nm in enumerate(names):
    mx, sm = (w[:, 5 + k] & 0xffffffff), (w[:, 5 + k] >> 32)
    print("%-18s wave-max: mean %6.1f p90 %5d p99 %5d max %5d | lane-mean %6.2f | in the slowest 1 %% of waves: wave-max mean %6.1f, lane-mean %6.2f"
          % (nm, mx.mean(), np.percentile(mx, 90), np.percentile(mx, 99), mx.max(), sm.mean() / 64.0, mx[heavy].mean(), sm[heavy].mean() / 64.0))
far = (w[:, 2] - w[:, 1]) > 20   # > 0.2 us spent in far rings
print("waves entering far rings: %.1f %%" % (100.0 * far.mean()))
start = (w[:, 0] - t0) / 100.0
end = (w[:, 4] - t0) / 100.0
print("slowest 1 %% of waves: start p10 %.1f p50 %.1f p90 %.1f us, end p50 %.1f max %.1f us; their share of all wave time %.1f %%"
      % (np.percentile(start[heavy], 10), np.percentile(start[heavy], 50), np.percentile(start[heavy], 90),
         np.percentile(end[heavy], 50), end[heavy].max(), 100.0 * dur[heavy].sum() / dur.sum()))
for lo, hi in ((0, 25), (25, 50), (50, 100), (100, 150), (150, 1e9)):
    sel = (end >= lo) & (end < hi)
    print("waves ending in [%g, %g) us: %5d" % (lo, hi, sel.sum()))
busy = np.zeros(int(end.max()) + 2)
for a, b in zip(start.astype(int), end.astype(int)):
    busy[a:b + 1] += 1
print("resident waves at t = 10/50/100/125/150/175 us:", [int(busy[min(t, len(busy) - 1)]) for t in (10, 50, 100, 125, 150, 175)])
print("wave start times: p50 %.1f us p90 %.1f us max %.1f us" % (np.percentile(start, 50), np.percentile(start, 90), start.max()))
