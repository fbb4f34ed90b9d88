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
import numpy as np


def run(fresh):
    times = []
    for k in range(105):
        if fresh:   # new records under the handles (outside the clock): their bounding boxes have to be measured again
            t.upload(tp)
            s.upload(sp)
        ctx.synchronize()
        t0 = time.perf_counter()
        ref.setInputSource(s)
        ref.setInputTarget(t)
        ref.align()
        ctx.synchronize()
        if k >= 5:
            times.append(time.perf_counter() - t0)
    return times


known = run(False)
times = run(True)
gi = ref.grid_info()
print("%s reference parameters, device clouds whose boxes an earlier load has measured: median %.3f ms per pair (p10 %.3f, p90 %.3f)" %
      (size, np.median(known) * 1e3, np.percentile(known, 10) * 1e3, np.percentile(known, 90) * 1e3))
print("%s reference parameters, device clouds: median %.3f ms per pair (p10 %.3f, p90 %.3f), %d correspondences, cell %.4f m" %
      (size, np.median(times) * 1e3, np.percentile(times, 10) * 1e3, np.percentile(times, 90) * 1e3, ref.result.n_correspondences, gi.cell_size))
