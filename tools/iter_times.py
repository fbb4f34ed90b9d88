#!/usr/bin/env python3
"""Dev probe: duration of every search launch of one bench pair (GPU only)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RSREG_DUMP_NN_MS"] = "1"
os.environ["RSREG_DIAG"] = "1"   # (per-launch times, wave stamps and dumps are the diagnostic build's: librsreg_diag.so, csrc/tunables.hpp)
import rsreg_amd  # noqa: E402,F401
from rsreg_amd import api, synth  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "N1M"
if "x" in size:
    size = tuple(int(v) for v in size.split("x"))   # WxH
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
pipeline = int(sys.argv[3]) if len(sys.argv) > 3 else 2
tgt, src = synth.render_frame(0, size, "bench"), synth.render_frame(1, size, "bench")
if os.environ.get("ITER_SHUFFLE"):   # both clouds in random order (an unorganized cloud: what the source's spatial sort is for)
    rng = np.random.default_rng(5)
    tgt = rsreg_amd.PointCloud(tgt.points[rng.permutation(len(tgt))], width=len(tgt), height=1, is_dense=False)
    src = rsreg_amd.PointCloud(src.points[rng.permutation(len(src))], width=len(src), height=1, is_dense=False)
guess = synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
icp = api.IterativeClosestPoint(api.Context(0, profiling=True))
icp.params = api.icp_params(max_iterations=iters, criteria_mode=1, pipeline_mode=pipeline, max_correspondence_distance=0.05)
icp.setInputSource(src)
icp.setInputTarget(tgt)
for _ in range(3):
    icp.align(guess)
r = icp.result
print("avg search launch %.1f us, between launches %.1f us, n_corr %d mse %.3e" %
      (r.ms_nn / r.n_nn_launches * 1e3, r.ms_reduce / max(r.n_nn_launches - 1, 1) * 1e3, r.n_correspondences, r.mse))
