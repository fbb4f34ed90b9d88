#!/usr/bin/env python3
"""Dev probe: wall time of the three registration schemes through the Python mirror (synchronous: the pipelined frame loops are
the C++ host layer's, tools/cpp_scheme_times.py), frames on the host in, merged cloud on the host out, device-resident frame loop
against host clouds (GPU only)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsreg_amd  # noqa: E402,F401
from rsreg_amd import schemes, synth  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "N300"
nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 16
frames = [synth.render_frame(k, size, "parity") for k in range(nframes)]


def copy_frames():
    return [rsreg_amd.PointCloud(f.points.copy(), width=f.width, height=f.height, is_dense=f.is_dense) for f in frames]


for name, make in (("IncrementalICP", lambda b: schemes.IncrementalICP(backend=b)),
                   ("ICPEdgeBasedRegistration", lambda b: schemes.ICPEdgeBasedRegistration(rads=-np.deg2rad(0.15), backend=b)),
                   ("NDTEdgeBasedRegistration", lambda b: schemes.NDTEdgeBasedRegistration(rads=-np.deg2rad(0.15), backend=b))):
    for bname, backend in (("device clouds", schemes.HipDeviceBackend), ("host clouds", schemes.HipBackend)):
        best = 1e9
        out = None
        for rep in range(3):
            out = None   # (the previous run's 157 MB go back outside the clock)
            fr = copy_frames()
            s = make(backend())
            t = time.perf_counter()
            out = s.registration(fr)
            best = min(best, time.perf_counter() - t)
        print("%-26s %-14s %2d x %s: %8.1f ms  (%.1f ms per frame), merged %d points" %
              (name, bname, nframes, size, best * 1e3, best * 1e3 / nframes, len(out)))
