#!/usr/bin/env python3
"""Folds the outputs of pcl_pin (oracle/_ref/out) and its inputs into tests/golden/pcl_pin.npz,
the fixture tests/test_pcl_pin.py looks for.  TEST INFRASTRUCTURE."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rsreg_amd  # noqa: E402


def read_matrix(path):
    with open(path) as f:
        rows, cols = (int(v) for v in f.readline().split())
        vals = np.array(f.read().split(), dtype=np.float64)
    return vals.reshape(rows, cols)


ref = os.path.join(ROOT, "oracle", "_ref")
out = {}
for name in sorted(os.listdir(os.path.join(ref, "out"))):
    p = os.path.join(ref, "out", name)
    key = os.path.splitext(name)[0]
    out[key] = read_matrix(p) if name.endswith(".txt") else rsreg_amd.load_pcd(p).points
    if name.endswith("binary_compressed.pcd"):   # the file itself, byte for byte: the LZF stream PCL wrote
        out[key + "_bytes"] = np.frombuffer(open(p, "rb").read(), np.uint8)
for name in ("pair-0", "pair-1", "chain-0", "chain-1", "chain-2", "chain-3"):
    out["in_" + name.replace("-", "")] = rsreg_amd.load_pcd(os.path.join(ref, "inputs", name + ".pcd")).points
out["in_guess"] = np.loadtxt(os.path.join(ref, "inputs", "guess.txt"))
out["in_rads"] = np.loadtxt(os.path.join(ref, "inputs", "rads.txt")).reshape(1)
for name in ("chain-0", "chain-1", "chain-2", "chain-3"):   # the frames are organized: keep their shape
    c = rsreg_amd.load_pcd(os.path.join(ref, "inputs", name + ".pcd"))
    out["in_" + name.replace("-", "") + "_shape"] = np.array([c.width, c.height])
dst = os.path.join(ROOT, "tests", "golden", "pcl_pin.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, "with", sorted(out))
