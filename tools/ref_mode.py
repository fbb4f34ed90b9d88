#!/usr/bin/env python3
"""Dev probe: the reference's own parameters (1 iteration, 1 cm gate) on a pair of device-resident frames (GPU only)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsreg_amd  # noqa: E402,F401
from rsreg_amd import api, synth  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "N1M"
ctx = api.default_context()
tp, sp = synth.render_frame(0, size, "parity"), synth.render_frame(1, size, "parity")
t, s = api.DeviceCloud(tp, ctx), api.DeviceCloud(sp, ctx)
ref = api.IterativeClosestPoint(ctx)
ref.params = api.icp_params(reference=True)
for k in range(25):
    if k == 5:
        ctx.synchronize()
        t0 = time.perf_counter()
    ref.setInputSource(s)
    ref.setInputTarget(t)
    ref.align()
ctx.synchronize()
gi = ref.grid_info()
print("%s reference parameters, device clouds: %.3f ms per pair, %d correspondences, cell %.4f m, %d cells occupied" %
      (size, (time.perf_counter() - t0) / 20 * 1e3, ref.result.n_correspondences, gi.cell_size, gi.n_cells))
