#!/usr/bin/env python3
"""Dev probe: one NDT alignment (the reference's parameters) of two voxel-filtered edge clouds on device clouds (GPU only)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsreg_amd  # noqa: E402,F401
from rsreg_amd import api, schemes, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
a, b = synth.render_frame(0, "N300", "parity"), synth.render_frame(1, "N300", "parity")
ea, eb = api.extract_edge_features(a), api.extract_edge_features(b)
vox = api.ApproximateVoxelGrid(api.default_context())
vox.setLeafSize(0.01, 0.01, 0.01)
vox.setInputCloud(api.DeviceCloud(ea)); ta = vox.filter()
vox.setInputCloud(api.DeviceCloud(eb)); sb = vox.filter()
sub = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if sub > 1:
    h = sb.download()
    pts = np.ascontiguousarray(h.points[::sub])
    sb = api.DeviceCloud(rsreg_amd.PointCloud(pts, width=len(pts), height=1, is_dense=h.is_dense))
ndt = schemes.HipBackend().ndt()
guess = schemes.rot_y(-np.deg2rad(0.15))
times = []
for k in range(reps + 5):
    api.default_context().synchronize()
    t = time.perf_counter()
    ndt.setInputSource(sb)
    ndt.setInputTarget(ta)
    ndt.align(guess)
    api.default_context().synchronize()
    if k >= 5:
        times.append(time.perf_counter() - t)
r = ndt.result
print("NDT align, %d -> %d points: median %.3f ms; iterations %d, passes %d, voxels %d" % (len(sb), len(ta), np.median(times) * 1e3, r.iterations, r.n_derivative_passes, r.n_voxels))
