#!/usr/bin/env python3
"""Dev probe: set source + set target + one reference-parameter alignment, a few times (under rocprofv3 --kernel-trace --stats:
the kernels of an index build and a source load with their durations)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsreg_amd  # noqa: E402,F401
from rsreg_amd import api, synth  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "N1M"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = api.default_context()
tp, sp = synth.render_frame(0, size, "parity"), synth.render_frame(1, size, "parity")
t, s = api.DeviceCloud(tp, ctx), api.DeviceCloud(sp, ctx)
ref = api.IterativeClosestPoint(ctx)
ref.params = api.icp_params(reference=True)
for _ in range(reps):
    ref.setInputSource(s)
    ref.setInputTarget(t)
    ref.align()
ctx.synchronize()
print("done", ref.result.n_correspondences)
