#!/usr/bin/env python3
"""Dev: the bench step (set source + set target + 30 iterations, N1M pair) one by one, each timed with a synchronisation behind it:
where the spread of bench.py's ms_per_step between processes comes from.  python tools/step_times.py [steps]"""
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

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
tgt, src = synth.render_frame(0, "N1M", "bench"), synth.render_frame(1, "N1M", "bench")
guess = synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
d_tgt = torch.from_numpy(tgt.points.view(np.uint8).reshape(-1)).cuda()
d_src = torch.from_numpy(src.points.view(np.uint8).reshape(-1)).cuda()
ctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
L = lib.lib()
prm = api.icp_params(max_iterations=30, criteria_mode=1, pipeline_mode=2, max_correspondence_distance=0.05)
g = np.ascontiguousarray(guess.T).copy()
res = lib.IcpResult()
stride = tgt.points.dtype.itemsize
ts = []
for k in range(steps):
    t0 = time.perf_counter()
    lib.check(L.rsreg_icp_set_source_device(ctx.h, d_src.data_ptr(), len(src), stride, 0), ctx.h)
    lib.check(L.rsreg_icp_set_target_device(ctx.h, d_tgt.data_ptr(), len(tgt), stride, 0, 0.05), ctx.h)
    lib.check(L.rsreg_icp_align(ctx.h, g.ctypes.data, C.byref(prm), C.byref(res), None, 0), ctx.h)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("steps (ms):", " ".join("%.2f" % t for t in ts))
print("median %.3f, mean of the last %d: %.3f" % (float(np.median(ts)), steps - 5, float(np.mean(ts[5:]))))
