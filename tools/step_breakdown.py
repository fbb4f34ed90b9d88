#!/usr/bin/env python3
"""Dev probe: where the host's time goes in one bench step (set source, set target, align), device-resident clouds."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsreg_amd  # noqa: E402,F401
from rsreg_amd import api, lib, synth  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "N1M"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
order = sys.argv[3] if len(sys.argv) > 3 else "source-first"
tgt, src = synth.render_frame(0, size, "bench"), synth.render_frame(1, size, "bench")
guess = synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
ctx = api.Context(0, profiling=False)
L = lib.lib()
stride = tgt.points.strides[0]
d_tgt = torch.from_numpy(tgt.points.view(np.uint8).reshape(-1)).cuda()
d_src = torch.from_numpy(src.points.view(np.uint8).reshape(-1)).cuda()
prm = api.icp_params(max_iterations=30, criteria_mode=1, pipeline_mode=2, max_correspondence_distance=0.05)
g = np.ascontiguousarray(guess.T).copy()
res = lib.IcpResult()
rows = []
for it in range(steps + 3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if order == "source-first":
        lib.check(L.rsreg_icp_set_source_device(ctx.h, d_src.data_ptr(), len(src.points), stride, 0), ctx.h)
        t1 = time.perf_counter()
        lib.check(L.rsreg_icp_set_target_device(ctx.h, d_tgt.data_ptr(), len(tgt.points), stride, 0, 0.05), ctx.h)
    else:
        lib.check(L.rsreg_icp_set_target_device(ctx.h, d_tgt.data_ptr(), len(tgt.points), stride, 0, 0.05), ctx.h)
        t1 = time.perf_counter()
        lib.check(L.rsreg_icp_set_source_device(ctx.h, d_src.data_ptr(), len(src.points), stride, 0), ctx.h)
    t2 = time.perf_counter()
    torch.cuda.synchronize()   # (the two loads are only queued: this is when the GPU is done with them)
    t3 = time.perf_counter()
    lib.check(L.rsreg_icp_align(ctx.h, g.ctypes.data, C.byref(prm), C.byref(res), None, 0), ctx.h)
    t4 = time.perf_counter()
    if it >= 3:
        rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
r = np.median(np.array(rows), axis=0) * 1e3
print("%s %s: first call returns after %.3f ms, second after %.3f ms, GPU done %.3f ms later, align %.3f ms; sum %.3f ms" %
      (size, order, r[0], r[1], r[2], r[3], r.sum()))
