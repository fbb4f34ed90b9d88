"""The context's helper threads (csrc/workers.hpp) under ThreadSanitizer and AddressSanitizer, with stub jobs (CPU only).

They carry the source load beside the index build (incremental_icp.hpp:57-58: setInputSource, setInputTarget), the frame
uploads / side jobs of the scheme loops (types.hpp:30-43) and the streamed download of the merged cloud."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "workers_tsan.cpp")


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_workers_under_sanitizer(san):
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "workers_" + san.split(",")[0])
        r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-pthread", SRC, "-o", exe],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout[-3000:]
        env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1")
        r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=env)
        assert r.returncode == 0 and "workers ok" in r.stdout, r.stdout[-4000:]
