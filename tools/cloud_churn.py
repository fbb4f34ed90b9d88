#!/usr/bin/env python3
"""Dev probe: what creating, copying and dropping device clouds costs (allocation churn of the frame loops)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsreg_amd  # noqa: E402,F401
from rsreg_amd import api, synth  # noqa: E402

ctx = api.default_context()
f = synth.render_frame(0, "N300", "parity")
big = api.DeviceCloud(f, ctx)
for _ in range(4):
    big = big + big          # 4.9 M points


def t(fn, reps=20):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        a = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        b = time.perf_counter()
        del r
        torch.cuda.synchronize()
        c = time.perf_counter()
        out.append((b - a, c - b))
    m = np.median(np.array(out), axis=0) * 1e3
    return m


small = api.DeviceCloud(f, ctx)
print("upload 307k frame: %.3f ms, drop %.3f ms" % tuple(t(lambda: api.DeviceCloud(f, ctx))))
print("copy 307k cloud:   %.3f ms, drop %.3f ms" % tuple(t(lambda: small.copy())))
print("copy 4.9M cloud:   %.3f ms, drop %.3f ms" % tuple(t(lambda: big.copy())))
print("4.9M + 307k:       %.3f ms, drop %.3f ms" % tuple(t(lambda: big + small)))
print("download 4.9M:     %.3f ms, drop %.3f ms" % tuple(t(lambda: big.download(), reps=8)))
print("download 307k:     %.3f ms, drop %.3f ms" % tuple(t(lambda: small.download())))
