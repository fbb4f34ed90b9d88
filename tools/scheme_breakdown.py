#!/usr/bin/env python3
"""Dev probe: where a scheme's wall time goes (every backend call timed with a device sync after it; GPU only)."""
import collections
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsreg_amd  # noqa: E402,F401
from rsreg_amd import api, schemes, synth  # noqa: E402
import torch  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "N300"
nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 16
which = sys.argv[3] if len(sys.argv) > 3 else "icp_edge"
frames = [synth.render_frame(k, size, "parity") for k in range(nframes)]
acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)


def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        acc[name] += time.perf_counter() - t
        cnt[name] += 1
        return r
    return w


class Timed(schemes.HipDeviceBackend):
    def __init__(self):
        super().__init__()
        for n in ("upload", "download", "transform", "edge_features", "concat"):
            setattr(self, n, timed(n, getattr(self, n)))

    def icp(self):
        o = super().icp()
        o.align = timed("icp.align (incl. set inputs)", o.align)
        return o

    def ndt(self):
        o = super().ndt()
        o.align = timed("ndt.align (incl. set inputs)", o.align)
        return o

    def voxel(self, leaf=None):
        o = super().voxel(leaf)
        o.filter = timed("voxel.filter", o.filter)
        return o


mk = {"incremental": lambda b: schemes.IncrementalICP(backend=b),
      "icp_edge": lambda b: schemes.ICPEdgeBasedRegistration(rads=-np.deg2rad(0.15), backend=b),
      "ndt_edge": lambda b: schemes.NDTEdgeBasedRegistration(rads=-np.deg2rad(0.15), backend=b)}[which]
for rep in range(2):
    acc.clear(); cnt.clear()
    fr = [rsreg_amd.PointCloud(f.points.copy(), width=f.width, height=f.height, is_dense=f.is_dense) for f in frames]
    s = mk(Timed())
    t = time.perf_counter()
    s.registration(fr)
    total = time.perf_counter() - t
print("%s, %d x %s: %.1f ms in all" % (which, nframes, size, total * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-32s %4d calls %8.2f ms  (%.2f ms each)" % (k, cnt[k], v * 1e3, v * 1e3 / cnt[k]))
print("  %-32s            %8.2f ms" % ("everything else (Python, model.append)", (total - sum(acc.values())) * 1e3))
