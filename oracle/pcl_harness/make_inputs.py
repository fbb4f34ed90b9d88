#!/usr/bin/env python3
"""Writes the inputs of the PCL pinning harness (oracle/pcl_harness/pcl_pin.cpp) under
oracle/_ref/inputs: the same seeded synthetic frames the parity tests use.  TEST INFRASTRUCTURE."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rsreg_amd  # noqa: E402
from rsreg_amd import synth  # noqa: E402

out = os.path.join(ROOT, "oracle", "_ref", "inputs")
os.makedirs(out, exist_ok=True)
os.makedirs(os.path.join(ROOT, "oracle", "_ref", "out"), exist_ok=True)
for k in (0, 1):
    rsreg_amd.save_pcd(os.path.join(out, "pair-%d.pcd" % k), synth.render_frame(k, "N300", "parity"))
for k in range(4):
    rsreg_amd.save_pcd(os.path.join(out, "chain-%d.pcd" % k), synth.render_frame(k, "50k", "parity"))
np.savetxt(os.path.join(out, "guess.txt"), synth.small_transform(0.1, (0.0, 0.0, 0.0)))
# the per-frame yaw the edge schemes are constructed with: the "parity" preset's 0.15 degrees per frame, negative like the
# reference's default (icp_edge_based_registration.hpp:133)
open(os.path.join(out, "rads.txt"), "w").write("%.9g\n" % -0.0026179939)
print("inputs written to", out)
