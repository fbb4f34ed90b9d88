#!/usr/bin/env python3
"""Dev probe: P independent pairs in flight on one GPU (one context and one host thread per pair; GPU only)."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsreg_amd  # noqa: E402,F401
from rsreg_amd import api, synth  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "N1M"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
tgt, src = synth.render_frame(0, size, "bench"), synth.render_frame(1, size, "bench")
guess = synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)


def worker(ctx, n, out, k):
    t, s = api.DeviceCloud(tgt, ctx), api.DeviceCloud(src, ctx)
    icp = api.IterativeClosestPoint(ctx)
    icp.params = api.icp_params(max_iterations=30, criteria_mode=1, pipeline_mode=2, max_correspondence_distance=0.05)
    for _ in range(n):
        icp.setInputSource(s)
        icp.setInputTarget(t)
        icp.align(guess)
    out[k] = icp.getFinalTransformation()


for P in (1, 2, 3, 4):
    ctxs = [api.Context(0) for _ in range(P)]
    out = [None] * P
    for c in ctxs:   # warm up (allocations)
        worker(c, 1, out, 0)
    th = [threading.Thread(target=worker, args=(ctxs[k], reps, out, k)) for k in range(P)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    if P == 1:
        ref = out[0].copy()
    same = [bool((o == ref).all()) for o in out]
    print("%d pair(s) in flight: %.3f ms per pair, %.3e point-pairs/s, same transform as alone: %s, max |diff| %.3g" %
          (P, dt / (P * reps) * 1e3, len(src) * 30 * P * reps / dt, same, max(float(np.abs(o - ref).max()) for o in out)))
