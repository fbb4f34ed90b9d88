"""The search has three interchangeable index paths: the dense cell-start table (default when
the grid fits), the brick hash (clouds whose box is too large for the table) and the LDS-staged
tile kernel over the brick hash (opt-in).  The parity suite must pass on each of them; the
alternative paths are selected per process, so they run in child processes."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env", [{"RSREG_FORCE_HASH": "1"}, {"RSREG_FORCE_HASH": "1", "RSREG_TILE": "1"},
                                 {"RSREG_DENSE_MAX_CELLS": "2000000"}],
                         ids=["brick-hash", "brick-hash+lds-tiles", "dense-only-when-small"])
def test_parity_suite_on_alternative_index_paths(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_icp_gpu.py"),
                        os.path.join(ROOT, "tests", "test_nn_fuzz_gpu.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
