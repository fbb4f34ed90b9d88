#!/usr/bin/env python3
"""Dev probe: the GPU voxel filter on one device-resident frame, repeated (for rocprofv3 --kernel-trace --stats)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsreg_amd  # noqa: E402,F401
from rsreg_amd import api, synth  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "N300"
leaf = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ctx = api.default_context()
f = api.DeviceCloud(synth.render_frame(1, size, "parity"), ctx)
v = api.ApproximateVoxelGrid(ctx)
v.setLeafSize(leaf, leaf, leaf)
v.setInputCloud(f)
for k in range(reps + 3):
    if k == 3:
        ctx.synchronize()
        t = time.perf_counter()
    out = v.filter()
ctx.synchronize()
print("%s leaf %g: %.3f ms per filter, %d points out" % (size, leaf, (time.perf_counter() - t) / reps * 1e3, len(out)))
