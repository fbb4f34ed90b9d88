#!/usr/bin/env python3
"""One-off: many random scenes, the fused search with every tile split (2 and 4 lanes per query) against the
unscheduled fused search and the staged kernels: identical sums and transforms (GPU only)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rsreg_amd  # noqa: E402
from rsreg_amd import api  # noqa: E402
from test_nn_fuzz_gpu import scene  # noqa: E402

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(77)
kinds = ["uniform", "plane", "clusters", "lattice", "line"]
gates = [0.004, 0.013, 0.05, 0.2, 1e30]
bad = 0
for it in range(n_scenes):
    kind = kinds[it % len(kinds)]
    nt, ns = int(rng.integers(2000, 60000)), int(rng.integers(2000, 40000))
    tgt = scene(rng, kind, nt).astype(np.float32)
    src = (scene(rng, kind, ns) + rng.uniform(-0.02, 0.02, 3)).astype(np.float32)
    if it % 3 == 0:
        tgt[rng.integers(0, nt, nt // 10)] = 0.0
        src[rng.integers(0, ns, ns // 10)] = 0.0
        src[rng.integers(0, ns, ns // 50)] = np.inf
    if it % 4 == 1:
        src[rng.integers(0, ns, ns // 8)] += rng.uniform(-1, 1, 3).astype(np.float32)
    gate = gates[int(rng.integers(0, len(gates)))]
    tc, sc = rsreg_amd.PointCloud.from_xyz(tgt), rsreg_amd.PointCloud.from_xyz(src)

    def run(pipeline, env):
        for k in list(os.environ):
            if k.startswith("RSREG_SCHED"):
                del os.environ[k]
        os.environ.update(env)
        icp = api.IterativeClosestPoint(api.Context(0))
        icp.params = api.icp_params(max_iterations=4, criteria_mode=1, pipeline_mode=pipeline, max_correspondence_distance=gate)
        icp.setInputSource(sc)
        icp.setInputTarget(tc)
        icp.align()
        r = icp.result
        return bytes(r.transform), bytes(r.sums_last), r.n_correspondences, r.state, icp.grid_info().index_kind

    base = run(2, {"RSREG_SCHED": "0"})
    staged = run(0, {"RSREG_SCHED": "0"})
    split = run(2, {"RSREG_SCHED_MIN_TILES": "1", "RSREG_SCHED_F2": "0.5", "RSREG_SCHED_F4": "0.5"})
    split2 = run(2, {"RSREG_SCHED_MIN_TILES": "1", "RSREG_SCHED_F2": "1.0", "RSREG_SCHED_F4": "0.0"})
    ok = base == staged == split == split2
    if not ok:
        bad += 1
    print("%2d %-8s nt %5d ns %5d gate %-7g index %d corr %6d : %s" % (it, kind, nt, ns, gate, base[4], base[2], "same" if ok else "DIFFERENT"))
print("scenes with differences:", bad)
sys.exit(1 if bad else 0)
