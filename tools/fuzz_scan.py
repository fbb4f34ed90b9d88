#!/usr/bin/env python3
"""One-off: many random scenes with a handful of source points; the index-free search (device-cloud target, at most 64
source points) against the indexed search (RSREG_NO_SCAN=1): identical matches, sums and transforms (GPU only)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rsreg_amd  # noqa: E402
from rsreg_amd import api  # noqa: E402
from test_nn_fuzz_gpu import scene  # noqa: E402

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(91)
kinds = ["uniform", "plane", "clusters", "lattice", "line"]
gates = [0.004, 0.013, 0.05, 0.2, None]
bad = 0
for it in range(n_scenes):
    kind = kinds[it % len(kinds)]
    nt, ns = int(rng.integers(32768, 400000)), int(rng.integers(1, 65))
    tgt = scene(rng, kind, nt).astype(np.float32)
    src = tgt[rng.integers(0, nt, ns)] + rng.normal(0, 0.01, (ns, 3)).astype(np.float32)
    if it % 3 == 0:
        tgt[rng.integers(0, nt, nt // 10)] = 0.0
        tgt[rng.integers(0, nt, nt // 50)] = np.nan
        src[rng.integers(0, ns, max(ns // 5, 1))] = 0.0
    if it % 4 == 1:
        src[rng.integers(0, ns, max(ns // 4, 1))] = tgt[rng.integers(0, nt, max(ns // 4, 1))]      # queries sitting on target points
        if ns > 3:
            src[0] = np.inf
            src[1] = src[2]
    gate = gates[int(rng.integers(0, len(gates)))]
    tc, sc = rsreg_amd.PointCloud.from_xyz(tgt), rsreg_amd.PointCloud.from_xyz(src)

    def run(no_scan):
        os.environ.pop("RSREG_NO_SCAN", None)
        if no_scan:
            os.environ["RSREG_NO_SCAN"] = "1"
        ctx = api.Context(0)
        icp = api.IterativeClosestPoint(ctx)
        kw = dict(max_iterations=3, criteria_mode=1)
        if gate is not None:
            kw["max_correspondence_distance"] = gate
        icp.params = api.icp_params(**kw)
        icp.setInputSource(api.DeviceCloud(sc, ctx))
        icp.setInputTarget(api.DeviceCloud(tc, ctx))
        icp.begin()
        idx, d2 = icp.search()
        icp.align()
        r = icp.result
        return icp.grid_info().index_kind, idx.tobytes(), d2.tobytes(), bytes(r.transform), bytes(r.sums_last), r.n_correspondences, r.state

    a, b = run(False), run(True)
    ok = a[0] == 2 and b[0] != 2 and a[1:] == b[1:]
    bad += not ok
    print("%2d %-8s nt %6d ns %2d gate %-7s corr %3d : %s" % (it, kind, nt, ns, gate, a[5], "same" if ok else "DIFFERENT"))
print("scenes with differences:", bad)
sys.exit(1 if bad else 0)
