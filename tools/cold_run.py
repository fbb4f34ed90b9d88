#!/usr/bin/env python3
"""Dev probe: the FIRST registration() of a process (the reference calls scheme->registration(clouds) once per process,
main.cpp:85,204-211) -- every scheme in fresh processes, run 0 against runs 1-2, with the steps of RSREG_SCHEME_COLD=1
(tests/cpp/scheme_runner.cpp) and the HIP runtime's own floor (tools/microbench/hip_floor.hip) beside it.  GPU only.
  python tools/cold_run.py [N300] [16] [processes]  ->  profiles/r06_cold_run.txt"""
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsreg_amd  # noqa: E402,F401
from rsreg_amd import cloud as cloud_io, synth  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "N300"
nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 16
procs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
out = os.path.join(ROOT, "tests", "cpp", "_build")
os.makedirs(out, exist_ok=True)
exe = os.path.join(out, "scheme_runner")
pkg = os.path.join(ROOT, "realsense-pointcloud_amd")
subprocess.run(["g++", "-std=c++17", "-O2", "-pthread", "-DRSREG_PCL_COMPAT_FAST_UNINIT", "-I", os.path.join(ROOT, "include"),
                os.path.join(ROOT, "tests", "cpp", "scheme_runner.cpp"), "-o", exe, "-L", pkg, "-lrsreg", "-Wl,-rpath," + pkg,
                "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"], check=True)
floor = os.path.join(out, "hip_floor")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "tools", "microbench", "hip_floor.hip"), "-o", floor], check=True)
for _ in range(procs):
    t0 = time.perf_counter()
    r = subprocess.run([floor], stdout=subprocess.PIPE, text=True, check=True)
    print("%s   [process wall %.1f ms]" % (r.stdout.strip(), (time.perf_counter() - t0) * 1e3))
with tempfile.TemporaryDirectory() as d:
    paths = []
    for k in range(nframes):
        p = os.path.join(d, "f%02d.pcd" % k)
        cloud_io.save_pcd(p, synth.render_frame(k, size, "bench"), binary=True)
        paths.append(p)
    if os.environ.get("COLD_TRACE"):
        # COLD_TRACE=<mode>: one cold process of that scheme under rocprofv3's HIP-API + kernel trace (gpurun_out/r06/cold_trace_<mode>/):
        # every runtime call of the first registration() with its duration
        mode = os.environ["COLD_TRACE"]
        tdir = os.path.join(ROOT, "gpurun_out", "r06", "cold_trace_" + mode)
        os.makedirs(tdir, exist_ok=True)
        env = dict(os.environ, RSREG_SCHEME_TIME="2", RSREG_SCHEME_FRAMES="1", RSREG_SCHEME_COLD="1")
        r = subprocess.run(["rocprofv3", "--hip-trace", "--kernel-trace", "--memory-copy-trace", "--output-format", "csv", "-d", tdir, "--",
                            exe, mode, os.path.join(d, "out_" + mode)] + paths, env=env, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
        print(r.stderr[-3000:])
        sys.exit(r.returncode)
    for mode in ("incremental", "icp_edge", "ndt_edge", "chain"):
        for cold in ("0", "1"):
            for p in range(procs if cold == "0" else 1):
                env = dict(os.environ, RSREG_SCHEME_TIME="3", RSREG_SCHEME_FRAMES="1", RSREG_SCHEME_COLD=cold)
                t0 = time.perf_counter()
                r = subprocess.run([exe, mode, os.path.join(d, "out_" + mode)] + paths, env=env, stderr=subprocess.PIPE, text=True, check=True)
                wall = (time.perf_counter() - t0) * 1e3
                tag = "steps taken apart" if cold == "1" else "as a caller sees it"
                for line in r.stderr.strip().splitlines():
                    if " run 0" in line or " cold:" in line or (p == 0 and cold == "0" and re.search(r" run [12]:", line)):
                        print("[%s, process %d] %s" % (tag, p, line))
