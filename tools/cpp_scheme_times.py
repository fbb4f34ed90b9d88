#!/usr/bin/env python3
"""Dev probe: the three schemes through the C++ host layer (include/rsreg/schemes.hpp, tests/cpp/scheme_runner.cpp):
frames on the host in, merged cloud on the host out, no Python in the loop.  GPU only."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsreg_amd  # noqa: E402,F401
from rsreg_amd import cloud as cloud_io, synth  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "N300"
nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 16
out = os.path.join(ROOT, "tests", "cpp", "_build")
os.makedirs(out, exist_ok=True)
exe = os.path.join(out, "scheme_runner")
pkg = os.path.join(ROOT, "realsense-pointcloud_amd")
subprocess.run(["g++", "-std=c++17", "-O2", "-pthread", "-DRSREG_PCL_COMPAT_FAST_UNINIT", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "scheme_runner.cpp"),
                "-o", exe, "-L", pkg, "-lrsreg", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"], check=True)
with tempfile.TemporaryDirectory() as d:
    paths = []
    for k in range(nframes):
        p = os.path.join(d, "f%02d.pcd" % k)
        cloud_io.save_pcd(p, synth.render_frame(k, size, "bench"), binary=True)
        paths.append(p)
    for mode in os.environ.get("RSREG_SCHEME_MODES", "incremental icp_edge ndt_edge").split():
        for host_loop in ("0", "1"):
            env = dict(os.environ, RSREG_SCHEME_TIME=os.environ.get("RSREG_SCHEME_REPS", "3") if host_loop == "0" else "2", RSREG_SCHEME_HOST_LOOP=host_loop)
            r = subprocess.run([exe, mode, os.path.join(d, "out_" + mode)] + paths, env=env, stderr=subprocess.PIPE, text=True, check=True)
            for line in r.stderr.strip().splitlines():
                print(("device clouds  " if host_loop == "0" else "host clouds    ") + line)
