#!/usr/bin/env python3
"""Dev check: every match of the last search launch of the 1 M bench pair against scipy's cKDTree (GPU + scipy)."""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rsreg_amd  # noqa: E402,F401
from rsreg_amd import api, synth  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "N1M"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
gate = 0.05
path = "/tmp/seed_verify.bin"
os.environ["RSREG_DUMP_SEED"] = path
tgt, src = synth.render_frame(0, size, "bench"), synth.render_frame(1, size, "bench")
guess = synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
icp = api.IterativeClosestPoint(api.Context(0))
icp.params = api.icp_params(max_iterations=iters, criteria_mode=1, pipeline_mode=2, max_correspondence_distance=gate)
icp.setInputSource(src)
icp.setInputTarget(tgt)
icp.align(guess)
n = int(icp.grid_info().n_source_distinct)
raw = np.fromfile(path, dtype=np.int32)
seed = raw[:n]
q = raw[n:5 * n].view(np.float32).reshape(n, 4)
P = raw[5 * n:].view(np.float32).reshape(-1, 4)
pts = np.stack([P[:, 0], P[:, 1], P[:, 3]], 1)
pidx = P[:, 2].view(np.int32)
valid = q[:, 3] != 0
tree = cKDTree(pts.astype(np.float64))
dist, nn = tree.query(q[:, :3].astype(np.float64), k=1)


def d2f(a, k):   # FLANN's float32 order
    dx, dy, dz = a[:, 0] - pts[k, 0], a[:, 1] - pts[k, 1], a[:, 2] - pts[k, 2]
    return (dx * dx + dy * dy) + dz * dz


inside = valid & (dist <= gate * 0.999)
have = seed >= 0
print("queries %d, with a target point inside the gate (cKDTree) %d, of these matched by the kernel %d" % (n, inside.sum(), (inside & have).sum()))
sel = np.nonzero(inside & have)[0]
dk = d2f(q[sel], seed[sel])
dt = d2f(q[sel], nn[sel])
worse = sel[dk > dt]          # the kernel's match is farther (in float32) than the tree's
tie_idx = sel[(dk == dt) & (pidx[seed[sel]] > pidx[nn[sel]])]
print("kernel match farther than the tree's: %d; equal distance but higher index than the tree's: %d (the tree has no index rule)" % (len(worse), len(tie_idx)))
missing = np.nonzero(inside & ~have)[0]
print("inside the gate by the tree but unmatched by the kernel: %d" % len(missing))
for i in list(worse[:5]) + list(missing[:5]):
    print("  query", i, "kernel", seed[i], "tree", nn[i], "tree dist", dist[i])
