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


@pytest.mark.parametrize("env", [{"RSREG_FORCE_HASH": "1"},
                                 {"RSREG_DENSE_MAX_CELLS": "2000000"},
                                 # round 6: the sort-based builds run on the library's own 64-bit radix sort and prefix sums
                                 # (csrc/osort.hpp, csrc/oscan.hpp; rocPRIM's until then)
                                 {"RSREG_COUNT_SORT": "0", "RSREG_KEYS64": "1"},
                                 {"RSREG_COUNT_SORT": "0", "RSREG_SCAN_APART": "1", "RSREG_FULL_TABLE": "1"}],
                         ids=["brick-hash", "dense-only-when-small", "sorted-build-64-bit-keys", "sorted-build-scans-apart-full-table"])
def test_parity_suite_on_alternative_index_paths(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_icp_gpu.py"),
                        os.path.join(ROOT, "tests", "test_nn_fuzz_gpu.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]


def test_ndt_between_icp_calls_leaves_the_hash_index_alone(tmp_path):
    """icp.setInputTarget once, then per frame ndt.align and icp.align without re-setting the ICP target (the
    usual PCL pattern): with the brick-hash index live, NDT's target build must not touch it."""
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
import oracle, rsreg_amd
from rsreg_amd import api, synth
tgt, src = synth.render_frame(0, "50k", "parity"), synth.render_frame(1, "50k", "parity")
ctx = api.Context(0)
icp = api.IterativeClosestPoint(ctx)
icp.params = api.icp_params(reference=True)
icp.setInputSource(src)
icp.setInputTarget(tgt)
icp.align()
assert icp.grid_info().index_kind == 0
T0 = icp.getFinalTransformation()
ndt = api.NormalDistributionsTransform(ctx)
ndt.params = api.ndt_params(reference=True)
ndt.setInputSource(src)
ndt.setInputTarget(tgt)
ndt.align()
icp.setInputSource(src)            # new source, SAME target object: the index is not rebuilt
icp.align()
assert (icp.getFinalTransformation() == T0).all()
o = oracle.IcpOracle()
o.set_target(tgt.points)
o.set_source(src.points)
r = o.align(None, oracle.IcpParams.reference())
assert np.linalg.norm(icp.getFinalTransformation() - r.T) < 1e-5 and icp.result.n_correspondences == r.n_correspondences
print("hash index intact")
''' % ROOT
    import subprocess, sys
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RSREG_FORCE_HASH="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0 and "hash index intact" in r.stdout, r.stderr[-3000:]
