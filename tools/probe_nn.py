#!/usr/bin/env python3
"""Dev probe: NN/fused kernel time vs iteration count, preset and cell cap (GPU only)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsreg_amd  # noqa: E402
from rsreg_amd import api, lib, synth  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "N1M"
preset = sys.argv[2] if len(sys.argv) > 2 else "bench"
gate = float(sys.argv[3]) if len(sys.argv) > 3 else (0.05 if preset == "bench" else 0.01)
its = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 2, 4, 8, 30]
pipeline = int(os.environ.get("PIPELINE", "1"))

tgt = synth.render_frame(0, size, preset)
src = synth.render_frame(1, size, preset)
guess = synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32) if preset == "bench" else None
ctx = api.Context(0, profiling=True)
icp = api.IterativeClosestPoint(ctx)
icp.setInputSource(src)
icp.setInputTarget(tgt)
prev_total, prev_n = 0.0, 0
for n_it in its:
    icp.params = api.icp_params(max_iterations=n_it, criteria_mode=1, pipeline_mode=pipeline, max_correspondence_distance=gate)
    icp._tgt_dirty = icp._tgt_dirty  # grid rebuilt only when the gate changes
    for rep in range(2):
        icp.align(guess)
    r = icp.result
    gi = icp.grid_info()
    marg = (r.ms_nn - prev_total) / max(r.n_nn_launches - prev_n, 1)
    print("iters %3d: nn total %.3f ms, avg %.4f ms, marginal avg of new iterations %.4f ms, reduce %.3f ms, corr %d, cell %.4f cells %d"
          % (n_it, r.ms_nn, r.ms_nn / r.n_nn_launches, marg, r.ms_reduce, r.n_correspondences, gi.cell_size, gi.n_cells), flush=True)
    prev_total, prev_n = r.ms_nn, r.n_nn_launches
