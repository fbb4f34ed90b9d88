#!/usr/bin/env python3
"""Dev probe: tools/cpp/pair_time.cpp (the reference-parameter pair as a C++ caller of the C ABI sees it) on synthetic frames."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsreg_amd  # noqa: E402,F401
from rsreg_amd import synth  # noqa: E402

out = os.path.join(ROOT, "tools", "_build")
os.makedirs(out, exist_ok=True)
exe = os.path.join(out, "pair_time")
pkg = os.path.join(ROOT, "realsense-pointcloud_amd")
subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "cpp", "pair_time.cpp"), "-o", exe,
                "-L", pkg, "-lrsreg", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"], check=True)
host = "--host" in sys.argv
for size in [a for a in sys.argv[1:] if not a.startswith("--")] or ["N1M", "N300", "50k"]:
    t, s = synth.render_frame(0, size, "parity"), synth.render_frame(1, size, "parity")
    with tempfile.TemporaryDirectory() as d:
        pt, ps = os.path.join(d, "t.f32"), os.path.join(d, "s.f32")
        t.points.tofile(pt)
        s.points.tofile(ps)
        subprocess.run([exe, pt, ps, str(len(t.points)), str(len(s.points))] + (["host"] if host else []), check=True)
