#!/usr/bin/env python3
"""Dev probe: wall time of each host-side call of one pair registration (1M, reference params)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsreg_amd  # noqa: E402
from rsreg_amd import api, lib, synth  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "N1M"
tgt, src = synth.render_frame(0, size, "parity"), synth.render_frame(1, size, "parity")
ctx = api.Context(0)
L = lib.lib()
prm = api.icp_params(reference=True)
res = lib.IcpResult()
out = src.points.copy()
n, stride = len(src), src.points.dtype.itemsize


def t(fn, reps=5):
    fn()
    a = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - a) / reps * 1e3


print("set_target(host)   %.3f ms" % t(lambda: lib.check(L.rsreg_icp_set_target(ctx.h, tgt.points.ctypes.data, n, stride, 0, 0.01), ctx.h)))
print("set_source(host)   %.3f ms" % t(lambda: lib.check(L.rsreg_icp_set_source(ctx.h, src.points.ctypes.data, n, stride, 0), ctx.h)))
print("align (no output)  %.3f ms" % t(lambda: lib.check(L.rsreg_icp_align(ctx.h, None, C.byref(prm), C.byref(res), None, 0), ctx.h)))
print("align (+aligned)   %.3f ms" % t(lambda: lib.check(L.rsreg_icp_align(ctx.h, None, C.byref(prm), C.byref(res), out.ctypes.data, stride), ctx.h)))
T = np.ascontiguousarray(np.eye(4, dtype=np.float32))
print("transform_cloud    %.3f ms" % t(lambda: lib.check(L.rsreg_transform_cloud(ctx.h, src.points.ctypes.data, out.ctypes.data, n, stride, 0, T.ctypes.data), ctx.h)))
print("numpy copy 32MB    %.3f ms" % t(lambda: src.points.copy()))

# ApproximateVoxelGrid: sequential host filter vs the GPU filter (same output), C calls on preallocated buffers
for size_v, leaf in (("N300", 0.01), ("N300", 1.0), ("N1M", 0.01)):
    c = synth.render_frame(2, size_v, "bench")
    pts = np.ascontiguousarray(c.points)
    outb = np.zeros_like(pts)
    outb[:] = pts          # touch every page once
    lf = np.array([leaf] * 3, np.float32)
    n_out = C.c_size_t(0)
    ms_h = t(lambda: lib.check(L.rsreg_approx_voxel_grid(pts.ctypes.data, len(pts), stride, lf.ctypes.data, outb.ctypes.data, C.byref(n_out))), reps=3)
    nh = n_out.value
    ms_g = t(lambda: lib.check(L.rsreg_approx_voxel_grid_gpu(ctx.h, pts.ctypes.data, len(pts), stride, lf.ctypes.data, outb.ctypes.data, C.byref(n_out)), ctx.h), reps=3)
    print("approx_voxel_grid %s leaf %.2f: host %.3f ms, gpu %.3f ms -> %d / %d points" % (size_v, leaf, ms_h, ms_g, nh, n_out.value))
