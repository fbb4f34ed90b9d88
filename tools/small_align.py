#!/usr/bin/env python3
"""Dev probe: one reference-parameter alignment of two edge-cloud-sized device clouds, repeated (GPU only)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsreg_amd  # noqa: E402,F401
from rsreg_amd import api, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
a, b = synth.render_frame(0, "N300", "parity"), synth.render_frame(1, "N300", "parity")
ea, eb = api.extract_edge_features(a), api.extract_edge_features(b)
vox = api.ApproximateVoxelGrid(api.default_context())
vox.setLeafSize(0.01, 0.01, 0.01)
da, db = api.DeviceCloud(ea), api.DeviceCloud(eb)
vox.setInputCloud(da); ta = vox.filter()
vox.setInputCloud(db); sb = vox.filter()
print("edge clouds: %d and %d points, filtered %d and %d" % (len(ea), len(eb), len(ta), len(sb)))
icp = api.IterativeClosestPoint()
icp.params = api.icp_params(reference=True)
if len(sys.argv) > 2:   # <iterations>: that many iterations whatever the criteria say, 5 cm gate
    icp.params = api.icp_params(max_iterations=int(sys.argv[2]), criteria_mode=1, max_correspondence_distance=0.05)
for k in range(reps + 5):
    if k == 5:
        t = time.perf_counter()
    icp.setInputSource(sb)
    icp.setInputTarget(ta)
    icp.align()
dt = (time.perf_counter() - t) / reps
print("align: %.3f ms, %d iteration(s), %d correspondences" % (dt * 1e3, icp.result.iterations, icp.result.n_correspondences))
