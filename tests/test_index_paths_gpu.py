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
                                 {"RSREG_COUNT_SORT": "0", "RSREG_SCAN_APART": "1", "RSREG_FULL_TABLE": "1"},
                                 # round 6 (end): the counting build leaves the occupancy words out for a small source whose gate fits into
                                 # ring 1 and the search reads them off the cell table (the default; the suite's 50 k clouds take it) --
                                 # here: the words built for every source, as until then
                                 {"RSREG_NO_NBR_FROM_TABLE": "1"}],
                         ids=["brick-hash", "dense-only-when-small", "sorted-build-64-bit-keys", "sorted-build-scans-apart-full-table",
                              "occupancy-words-for-small-sources-too"])
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


def test_search_without_occupancy_words_gives_the_matches_of_the_search_with_them(tmp_path):
    """The same pairs with the occupancy words built (RSREG_NO_NBR_FROM_TABLE=1) and read off the table (default for a source of
    at most 65 536 points and a gate inside ring 1): every match index and squared distance, the sums and the transform bit for
    bit -- on a dense pair, on a sparse edge-like pair (lone cells: most neighbours empty) and with queries outside the target's
    box (border cells of the padded table)."""
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
import rsreg_amd
from rsreg_amd import api, synth
import rsreg_amd as rs
out = []
def pair(src, tgt, gate, guess=None):
    icp = api.IterativeClosestPoint(api.Context(0))
    icp.setMaxCorrespondenceDistance(gate)
    icp.setMaximumIterations(3)
    icp.setTransformationEpsilon(0)
    icp.setEuclideanFitnessEpsilon(0)
    icp.setInputSource(src)
    icp.setInputTarget(tgt)
    al = icp.align(guess)
    r = icp.result
    out.append((bytes(r.transform), r.n_correspondences, r.iterations, np.stack([al.points[k] for k in "xyz"]).tobytes(), bytes(memoryview(np.array(r.sums_last)))))
a, b = synth.render_frame(0, "50k", "parity"), synth.render_frame(1, "50k", "parity")
pair(b, a, 0.01)
pair(b, a, 0.02, synth.small_transform(1.0, (0.01, -0.02, 0.015)).astype(np.float32))
e0 = rs.PointCloud(np.ascontiguousarray(a.points[::7]))      # sparse: lone cells
e1 = rs.PointCloud(np.ascontiguousarray(b.points[3::11]))
pair(e1, e0, 0.01)
far = rs.PointCloud(b.points.copy())
far.points["x"] += 0.8                                        # half of the queries outside the target's box
far.points["z"] -= 0.5
pair(far, a, 0.015)
import hashlib
print("digest", hashlib.sha256(b"".join(x[0] + x[3] + x[4] + bytes([x[2]]) + x[1].to_bytes(8, "little") for x in out)).hexdigest(), [x[1] for x in out])
''' % ROOT
    res = []
    for words in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RSREG_NO_NBR_FROM_TABLE=words), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=600)
        assert r.returncode == 0 and "digest" in r.stdout, r.stderr[-3000:]
        res.append([l for l in r.stdout.splitlines() if l.startswith("digest")][0])
    assert res[0] == res[1], res
    assert "[0, 0, 0, 0]" not in res[0]
