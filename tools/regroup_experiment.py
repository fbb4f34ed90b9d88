#!/usr/bin/env python3
"""Dev experiment (timing only, no library change): what would waves of like-cost queries buy the search launch?

The engine keeps a source in the caller's order when told to (RSREG_PLAIN_SOURCE_MAX), so the ORDER of the queries can be
chosen from outside: (A) the distinct source points in Morton order (what the library's own sort gives), and (B) the same
points re-dealt inside windows of W consecutive queries by descending cost, the cost being each query's own step count
(own cell + ring 1 + far) in the last of ten iterations of (A), read from the diagnostic build's per-lane counters.  (B)
is the best any per-window ordering by cost classes could do (every class its own).  GPU only."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RSREG_DIAG"] = "1"
os.environ["RSREG_PLAIN_SOURCE_MAX"] = "4000000"
import rsreg_amd  # noqa: E402
from rsreg_amd import api, synth  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "N1M"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tgt, src = synth.render_frame(0, size, "bench"), synth.render_frame(1, size, "bench")
guess = synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)

# the distinct finite points of the source (the library merges exact copies: one query each), in Morton order of 1.6 cm cells
xyz = np.stack([src.points[c] for c in "xyz"], axis=1)
ok = np.isfinite(xyz).all(axis=1)
_, first = np.unique(xyz[ok].view([("", xyz.dtype)] * 3), return_index=True)
idx = np.flatnonzero(ok)[np.sort(first)]
q = np.floor((xyz[idx] - xyz[idx].min(axis=0)) / 0.016).astype(np.uint64)


def spread(v):
    v = v & np.uint64(0x1fffff)
    v = (v | (v << np.uint64(32))) & np.uint64(0x1f00000000ffff)
    v = (v | (v << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
    v = (v | (v << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
    v = (v | (v << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
    return v


morton = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))
order = idx[np.argsort(morton, kind="stable")]


def cloud(ix):
    return rsreg_amd.PointCloud(src.points[ix].copy(), width=len(ix), height=1, is_dense=False)


def run(ix, diag_path=None):
    if diag_path:
        os.environ["RSREG_WAVE_TIMES"] = diag_path
    else:
        os.environ.pop("RSREG_WAVE_TIMES", None)
    icp = api.IterativeClosestPoint(api.Context(0, profiling=True))   # (a new context reads the switches again)
    icp.params = api.icp_params(max_iterations=10 if diag_path else iters, criteria_mode=1, pipeline_mode=2, max_correspondence_distance=0.05)
    icp.setInputSource(cloud(ix))
    icp.setInputTarget(tgt)
    for _ in range(1 if diag_path else 3):
        icp.align(guess)
    r = icp.result
    return r.ms_nn / r.n_nn_launches * 1e3, r.n_correspondences, r.mse


path = os.path.join(tempfile.gettempdir(), "rsreg_regroup_wave_times.bin")
run(order, path)
raw = np.fromfile(path, dtype=np.uint64)
n = len(order)
nw = (n + 63) // 64
lane = raw[16 * nw:].view(np.uint32)[:n]
own, r1, frow, fscan = (lane & 255).astype(np.int64), (lane >> 8 & 255).astype(np.int64), (lane >> 16 & 255).astype(np.int64), (lane >> 24).astype(np.int64)
cost = own + r1 + 2 * frow + fscan


def wave_max_sum(c):
    pad = (-len(c)) % 64
    return int(np.concatenate([c, np.zeros(pad, c.dtype)]).reshape(-1, 64).max(axis=1).sum())


print("%d distinct queries; steps per query: mean %.2f, p50 %d p90 %d p99 %d max %d; sum over waves of the slowest lane: %d (sum of lanes / 64: %d)"
      % (n, cost.mean(), *[int(np.percentile(cost, p)) for p in (50, 90, 99, 100)], wave_max_sum(cost), cost.sum() // 64))
us, nc, mse = run(order)
print("Morton order (the library's own order, fed from outside):              avg search launch %6.1f us   n_corr %d mse %.4e" % (us, nc, mse))
for win in (128, 512, 1024, 4096):
    key = (np.arange(n) // win) * 100000 - np.minimum(cost, 99999)
    o = np.argsort(key, kind="stable")
    us, nc, mse = run(order[o])
    print("re-dealt by descending cost inside windows of %4d: wave-steps x %.2f -> avg search launch %6.1f us   n_corr %d mse %.4e"
          % (win, wave_max_sum(cost) / wave_max_sum(cost[o]), us, nc, mse))
