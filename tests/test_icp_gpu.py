"""GPU parity tests of the ICP path, through the C ABI (librsreg.so), against
  (a) the independent numpy/scipy golden vectors (tests/golden), and
  (b) the CPU oracle (oracle/*.c) on the same seeded synthetic inputs.
Bars: nearest-neighbour indices and float32 squared distances bit-exact; the 17 f64 sums to
1e-11 relative; 4x4 transforms to 2e-6 per element after one iteration, and within the
north-star tolerance of 1e-4 Frobenius (we assert 2e-5) after multi-iteration runs.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

T_TOL = 2e-6


@pytest.fixture(scope="module")
def api(rs):
    from rsreg_amd import api as a, lib
    lib.build()
    if a.device_count() < 1:
        pytest.fail("no HIP device: the product has no CPU fallback")
    return a


def _xyz(p):
    return np.stack([p["x"], p["y"], p["z"]], 1)


def _ref_icp(api, src, tgt, **kw):
    icp = api.IterativeClosestPoint()
    icp.params = api.icp_params(reference=True, **kw)
    icp.setInputSource(src)
    icp.setInputTarget(tgt)
    return icp


@pytest.mark.parametrize("case", ["kat_exact", "crop_parity"])
def test_golden_stagewise_reference_params(api, rs, golden, case):
    g = golden(case)
    icp = _ref_icp(api, rs.PointCloud(g["src"].copy()), rs.PointCloud(g["tgt"].copy()))
    icp.begin(g["guess"])
    idx, d2 = icp.search()
    gi, gd = g["ref_it0_index"], g["ref_it0_sqr_dist"]
    np.testing.assert_array_equal(idx, gi)           # lowest-index tie-break: exact
    np.testing.assert_array_equal(d2[gi >= 0], gd[gi >= 0])
    s = icp.sums()
    np.testing.assert_allclose(s, g["ref_it0_sums"], rtol=1e-12, atol=1e-12)
    t_inc, done = icp.update(s)
    np.testing.assert_allclose(t_inc, g["ref_it0_t_inc"], atol=T_TOL)
    res = icp.end()
    assert done and (res.iterations, res.state, res.converged) == tuple(g["ref_meta"])
    np.testing.assert_allclose(api._rowmajor(res.transform), g["ref_final"], atol=T_TOL)


def test_kat_recovers_known_transform(api, rs, golden):
    g = golden("kat_exact")
    icp = _ref_icp(api, rs.PointCloud(g["src"].copy()), rs.PointCloud(g["tgt"].copy()))
    out = icp.align()
    assert icp.hasConverged() and icp.result.iterations == 1
    assert np.abs(icp.getFinalTransformation() - g["T_true"]).max() < 2e-6
    assert np.abs(out.xyz - _xyz(g["tgt"])).max() < 1e-4
    assert (out.points["w"] == 1).all()
    np.testing.assert_array_equal(out.points["rgba"], g["src"]["rgba"])


@pytest.mark.parametrize("pipeline", [0, 1, 2])
def test_golden_fixed_iterations_and_pcl_criteria(api, rs, golden, pipeline):
    g = golden("crop_parity")
    icp = api.IterativeClosestPoint()
    icp.params = api.icp_params(max_iterations=8, criteria_mode=1, pipeline_mode=pipeline,
                                max_correspondence_distance=0.02, transformation_epsilon=1e-12,
                                euclidean_fitness_epsilon=1e-12)
    icp.setInputSource(rs.PointCloud(g["src"].copy()))
    icp.setInputTarget(rs.PointCloud(g["tgt"].copy()))
    icp.align(g["guess"])
    r = icp.result
    assert (r.iterations, r.state, r.converged) == tuple(g["fixed8_meta"])
    np.testing.assert_allclose(icp.getFinalTransformation(), g["fixed8_final"], atol=5e-6)
    np.testing.assert_allclose(r.mse, g["fixed8_mse"][0], rtol=1e-4)

    g = golden("crop_bench")
    icp = api.IterativeClosestPoint()
    icp.params = api.icp_params(max_iterations=30, pipeline_mode=pipeline, max_correspondence_distance=0.05,
                                transformation_epsilon=1e-9, euclidean_fitness_epsilon=1e-7)
    icp.setInputSource(rs.PointCloud(g["src"].copy()))
    icp.setInputTarget(rs.PointCloud(g["tgt"].copy()))
    icp.begin(g["guess"])
    idx, d2 = icp.search()
    np.testing.assert_array_equal(idx, g["pcl_it0_index"])
    icp.end()
    icp.align(g["guess"])
    r = icp.result
    assert (r.iterations, r.state, r.converged) == tuple(g["pcl_meta"])
    np.testing.assert_allclose(icp.getFinalTransformation(), g["pcl_final"], atol=2e-5)


@pytest.fixture(scope="module")
def frames(rs):
    s = rs.synth
    return {
        ("50k", "parity"): (s.render_frame(1, "50k", "parity"), s.render_frame(0, "50k", "parity")),
        ("N300", "parity"): (s.render_frame(1, "N300", "parity"), s.render_frame(0, "N300", "parity")),
        ("N300", "bench"): (s.render_frame(1, "N300", "bench"), s.render_frame(0, "N300", "bench")),
    }


@pytest.mark.parametrize("size,dedup", [("50k", True), ("50k", False), ("N300", True)])
def test_vs_oracle_reference_params(api, orc, frames, size, dedup):
    """(dedup: whether the checker's kd-tree drops the exact copies among the target's points -- the (0,0,0) pixels, a
    tenth of a frame -- before it is built.  PCL keeps them; with the lowest index winning among equidistant points the
    matches are the same either way, and the engine, which always drops them, must agree with both.)"""
    src, tgt = frames[(size, "parity")]
    icp = _ref_icp(api, src, tgt)
    icp.begin()
    idx, d2 = icp.search()
    o = orc.IcpOracle()
    o.set_target(tgt.points, dedup=dedup)
    o.set_source(src.points)
    p = orc.IcpParams.reference()
    o.begin(None, p)
    oi, od = o.search()
    np.testing.assert_array_equal(idx, oi)
    np.testing.assert_array_equal(d2[oi >= 0], od[oi >= 0])
    s, so = icp.sums(), o.sums()
    np.testing.assert_allclose(s, so, rtol=1e-11, atol=1e-11)
    assert s[0] == so[0] > 0.5 * len(src)
    t_inc, done = icp.update(s)
    to, _ = o.update(so)
    np.testing.assert_allclose(t_inc, to, atol=T_TOL)
    res, aligned = icp.end(want_aligned=True)
    ro, oaligned = o.end(want_aligned=True)
    assert (res.iterations, res.state, res.converged) == (ro.iterations, ro.state, ro.converged) == (1, 2, 1)
    np.testing.assert_allclose(api._rowmajor(res.transform), ro.T, atol=T_TOL)
    np.testing.assert_allclose(aligned, oaligned, atol=1e-5)
    gi = icp.grid_info()
    assert gi.n_target_points == len(tgt) and gi.n_unique_points < gi.n_target_points  # (0,0,0) pixels collapse


@pytest.mark.parametrize("pipeline", [0, 1, 2])
def test_vs_oracle_bench_mode_30_iterations(api, orc, frames, rs, pipeline):
    """BASELINE configs[1] shape: 2 x 300k, 30 fixed iterations, 5 cm gate."""
    src, tgt = frames[("N300", "bench")]
    guess = rs.synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
    icp = api.IterativeClosestPoint()
    icp.params = api.icp_params(max_iterations=30, criteria_mode=1, pipeline_mode=pipeline,
                                max_correspondence_distance=0.05)
    icp.setInputSource(src)
    icp.setInputTarget(tgt)
    icp.align(guess)
    o = orc.IcpOracle()
    o.set_target(tgt.points, dedup=True)
    o.set_source(src.points)
    p = orc.IcpParams.default()
    p.max_iterations, p.criteria_mode, p.max_correspondence_distance, p.num_threads = 30, 1, 0.05, 8
    ro = o.align(guess, p)
    r = icp.result
    assert (r.iterations, r.state, r.converged) == (ro.iterations, ro.state, ro.converged) == (30, 1, 1)
    err = np.linalg.norm(icp.getFinalTransformation() - ro.T)
    assert err < 2e-5, err                              # north-star bar: 1e-4 Frobenius
    assert abs(r.n_correspondences - ro.n_correspondences) <= 1e-4 * ro.n_correspondences
    gt = rs.synth.ground_truth(1, 0, "bench")
    assert np.linalg.norm(icp.getFinalTransformation() - gt) < np.linalg.norm(guess - gt)


def test_pipelines_are_bit_identical(api, frames):
    """Staged kernels, the fused kernel, and the device-resident loop (3x3 solve on the GPU) add
    the same numbers in the same order and run the same f64 solve: identical bits."""
    src, tgt = frames[("50k", "parity")]
    for gate, iters in ((0.02, 6), (0.01, 4)):
        out = []
        for pipeline in (0, 1, 2):
            icp = api.IterativeClosestPoint()
            icp.params = api.icp_params(max_iterations=iters, criteria_mode=1, pipeline_mode=pipeline,
                                        max_correspondence_distance=gate)
            icp.setInputSource(src)
            icp.setInputTarget(tgt)
            icp.align()
            r = icp.result
            out.append((bytes(r.transform), bytes(r.sums_last), r.n_correspondences, r.iterations, r.state, r.converged, r.mse))
        assert out[0] == out[1] == out[2], gate


def test_device_loop_edge_cases(api, rs, frames):
    src, tgt = frames[("50k", "parity")]
    # (a) clouds too far apart: no correspondences in the first iteration -> same outcome as the host loop
    far = rs.synth.small_transform(0.0, (5.0, 0.0, 0.0)).astype(np.float32)
    out = []
    for pipeline in (1, 2):
        icp = api.IterativeClosestPoint()
        icp.params = api.icp_params(max_iterations=5, criteria_mode=1, pipeline_mode=pipeline, max_correspondence_distance=0.01)
        icp.setInputSource(src)
        icp.setInputTarget(tgt)
        icp.align(far)
        r = icp.result
        out.append((bytes(r.transform), r.iterations, r.state, r.converged, r.n_correspondences))
    assert out[0] == out[1]
    assert out[0][1:4] == (0, 5, 0)          # RSREG_CONV_NO_CORRESPONDENCES, not converged
    # (b) with PCL's criteria the mode falls back to the host loop (the stop decision is the host's)
    out = []
    for pipeline in (1, 2):
        icp = api.IterativeClosestPoint()
        icp.params = api.icp_params(reference=True)
        icp.params.pipeline_mode = pipeline
        icp.setInputSource(src)
        icp.setInputTarget(tgt)
        icp.align()
        r = icp.result
        out.append((bytes(r.transform), r.iterations, r.state, r.converged))
    assert out[0] == out[1] and out[0][1] == 1
    # (c) the aligned cloud and max_iterations = 1
    icp = api.IterativeClosestPoint()
    icp.params = api.icp_params(max_iterations=1, criteria_mode=1, pipeline_mode=2, max_correspondence_distance=0.01)
    icp.setInputSource(src)
    icp.setInputTarget(tgt)
    aligned = icp.align()
    T = icp.getFinalTransformation().astype(np.float64)
    xyz = np.stack([src.points["x"], src.points["y"], src.points["z"]], 1).astype(np.float64)
    want = xyz @ T[:3, :3].T + T[:3, 3]
    got = np.stack([aligned.points["x"], aligned.points["y"], aligned.points["z"]], 1)
    np.testing.assert_allclose(got, want, atol=1e-5)


def test_run_to_run_determinism(api, frames):
    src, tgt = frames[("N300", "parity")]
    res = []
    for _ in range(2):
        icp = _ref_icp(api, src, tgt, max_iterations=4, criteria_mode=1)
        icp.align()
        res.append((bytes(icp.result.transform), bytes(icp.result.sums_last)))
    assert res[0] == res[1]


def test_source_block_sums_add_up(api, frames):
    """What the N-GPU path relies on: the sums of source blocks add to the sums of the whole."""
    src, tgt = frames[("50k", "parity")]
    whole = _ref_icp(api, src, tgt)
    whole.begin()
    whole.search()
    s_all = whole.sums()
    whole.end()
    n = len(src)
    acc = np.zeros(17)
    for lo, hi in ((0, n // 3), (n // 3, n // 2), (n // 2, n)):
        part = _ref_icp(api, src.points[lo:hi], tgt)
        part.begin()
        part.search()
        acc += part.sums()
        part.end()
    np.testing.assert_allclose(acc, s_all, rtol=1e-12, atol=1e-9)
    assert acc[0] == s_all[0]


def test_edge_cases(api, rs, golden):
    g = golden("kat_exact")
    tgt = rs.PointCloud(g["tgt"].copy())
    src = rs.PointCloud(g["src"].copy())
    # far apart: fewer than 3 correspondences -> not converged, identity, state NO_CORRESPONDENCES
    far = src.copy()
    far.points["x"] += 50.0
    icp = _ref_icp(api, far, tgt)
    icp.align()
    assert not icp.hasConverged() and icp.getConvergenceState() == "NO_CORRESPONDENCES" and icp.result.iterations == 0
    np.testing.assert_array_equal(icp.getFinalTransformation(), np.eye(4, dtype=np.float32))
    # non-finite points on both sides are skipped
    s2, t2 = src.copy(), tgt.copy()
    s2.points["x"][5] = np.nan
    t2.points["y"][7] = np.inf
    icp = _ref_icp(api, s2, t2)
    icp.begin()
    idx, _ = icp.search()
    assert idx[5] == -1 and idx[7] == -1 and (idx[[4, 6, 8]] == [4, 6, 8]).all()
    icp.end()
    icp.align()
    assert np.abs(icp.getFinalTransformation() - g["T_true"]).max() < 2e-6
    # empty source / empty target / all-invalid target
    for s_, t_ in ((rs.PointCloud(src.points[:0].copy()), tgt), (src, rs.PointCloud(tgt.points[:0].copy()))):
        icp = _ref_icp(api, s_, t_)
        icp.align()
        assert not icp.hasConverged() and icp.getConvergenceState() == "NO_CORRESPONDENCES"
    t3 = tgt.copy()
    t3.points["z"] = np.nan
    icp = _ref_icp(api, src, t3)
    icp.align()
    assert not icp.hasConverged()
    # max_iterations = 0 edge: PCL still runs one iteration, then reports ITERATIONS
    icp = _ref_icp(api, src, tgt, max_iterations=0, transformation_epsilon=0.0, euclidean_fitness_epsilon=-1.0)
    icp.align()
    assert icp.result.iterations == 1 and icp.getConvergenceState() == "ITERATIONS"


def test_guess_is_applied_and_composed(api, rs, golden, orc):
    g = golden("kat_exact")
    G = rs.synth.small_transform(0.05, (0.001, 0.0, -0.0005)).astype(np.float32)
    icp = _ref_icp(api, rs.PointCloud(g["src"].copy()), rs.PointCloud(g["tgt"].copy()))
    icp.align(G)
    o = orc.IcpOracle()
    o.set_target(g["tgt"])
    o.set_source(g["src"])
    r = o.align(G, orc.IcpParams.reference())
    np.testing.assert_allclose(icp.getFinalTransformation(), r.T, atol=T_TOL)
    assert np.abs(icp.getFinalTransformation() - g["T_true"]).max() < 3e-6


def test_unbounded_gate_matches_brute_force(api, rs, orc):
    """PCL's default gate is sqrt(DBL_MAX): the ring search must stay exact without one."""
    rng = np.random.default_rng(3)
    tgt = (rng.random((3000, 3)).astype(np.float32) * 2 - 1)
    src = (rng.random((2000, 3)).astype(np.float32) * 2.6 - 1.3)
    icp = api.IterativeClosestPoint()
    icp.params = api.icp_params(max_iterations=3, criteria_mode=1)
    icp.setInputSource(np.ascontiguousarray(src))
    icp.setInputTarget(np.ascontiguousarray(tgt))
    icp.begin()
    idx, d2 = icp.search()
    icp.end()
    o = orc.IcpOracle()
    o.set_target(tgt)
    o.set_source(src)
    p = orc.IcpParams.default()
    p.nn_mode = 1
    o.begin(None, p)
    oi, od = o.search()
    np.testing.assert_array_equal(idx, oi)
    np.testing.assert_array_equal(d2, od)


def test_transform_point_cloud(api, rs, orc, golden):
    g = golden("crop_parity")
    T = g["ref_final"]
    c = rs.PointCloud(g["src"].copy(), is_dense=False)
    c.points["x"][3] = np.nan
    out = api.transformPointCloud(c, T)
    exp = orc.transform_cloud(c.points, T, is_dense=False)
    for f in ("x", "y", "z", "w", "rgba"):
        np.testing.assert_array_equal(out.points[f], exp[f])
    assert (out.width, out.height) == (c.width, c.height)


@pytest.mark.parametrize("size", ["N1M"])
def test_full_size_properties(api, rs, size):
    """BASELINE full size (1M <-> 1M): size-independent properties, no oracle needed."""
    tgt = rs.synth.render_frame(0, size, "parity")
    # (1) a cloud registered against itself: every finite point matches at distance 0, T = I
    icp = _ref_icp(api, tgt, tgt)
    icp.begin()
    idx, d2 = icp.search()
    s = icp.sums()
    icp.end()
    assert (d2 == 0).all() and (idx >= 0).all() and s[0] == len(tgt) and s[16] == 0
    xyz = tgt.xyz
    np.testing.assert_array_equal(xyz[idx], xyz)            # matched coordinates are the point itself
    nz = tgt.points["z"] != 0
    np.testing.assert_array_equal(idx[nz], np.nonzero(nz)[0])  # unique points match themselves
    assert (idx[~nz] == np.nonzero(~nz)[0][0]).all()        # duplicates -> lowest index copy
    np.testing.assert_allclose(api.umeyama_from_sums(s), np.eye(4), atol=1e-6)
    # (2) exactly moved copy of the valid points: one iteration recovers the motion
    T = rs.synth.small_transform(0.005, (0.0002, -0.0001, 0.00015))
    base = rs.PointCloud(np.ascontiguousarray(tgt.points[nz]))
    moved = base.copy()
    p64 = base.xyz.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    moved.points["x"], moved.points["y"], moved.points["z"] = p64[:, 0], p64[:, 1], p64[:, 2]
    icp = _ref_icp(api, base, moved)
    icp.begin()
    idx, _ = icp.search()
    icp.end()
    assert (idx == np.arange(len(base))).mean() > 0.999
    icp.align()
    assert icp.hasConverged() and icp.result.iterations == 1
    assert np.abs(icp.getFinalTransformation() - T).max() < 2e-5


def test_source_load_on_the_worker_thread_gives_the_same_bits(rs):
    """The host side of a source load runs on a thread of the context's own (rsreg_ctx.hpp: SourceWorker) so that the
    caller's thread goes straight on to the target's index build; RSREG_NO_WORKER=1 keeps it on the caller's thread.
    Same transform, same correspondences, bit for bit, in either call order, with host clouds and with cloud handles --
    and a context is torn down cleanly while a load is still in flight."""
    import subprocess
    import sys
    code = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
import rsreg_amd
from rsreg_amd import api, synth
tgt, src = synth.render_frame(0, "50k", "bench"), synth.render_frame(1, "50k", "bench")
h = hashlib.sha256()
for order in ("source-first", "target-first"):
    for dev in (False, True):
        ctx = api.Context(0)
        icp = api.IterativeClosestPoint(ctx)
        icp.params = api.icp_params(max_iterations=5, criteria_mode=1, max_correspondence_distance=0.05)
        s, t = (api.DeviceCloud(src, ctx), api.DeviceCloud(tgt, ctx)) if dev else (src, tgt)
        if order == "source-first":
            icp.setInputSource(s); icp.setInputTarget(t)
        else:
            icp.setInputTarget(t); icp.setInputSource(s)
        for _ in range(3):                       # repeated loads: a job is posted while the last one may still be running
            icp.setInputSource(s)
        icp.align()
        h.update(icp.getFinalTransformation().tobytes())
        h.update(np.int64(icp.result.n_correspondences).tobytes())
ctx = api.Context(0)                             # torn down with a load in flight
icp = api.IterativeClosestPoint(ctx)
icp.setInputSource(src)
icp.setInputTarget(tgt)
icp._sync_inputs()                           # source queued on the worker, target built: no align
del icp, ctx
print(h.hexdigest())
''' % ROOT
    out = {}
    for no_worker in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RSREG_NO_WORKER=no_worker), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[no_worker] = r.stdout.strip().splitlines()[-1]
    assert out["0"] == out["1"] and len(out["0"]) == 64


@pytest.mark.gpu
def test_sources_of_every_size_through_one_context(rs):
    """The source load takes different paths by size -- the caller's order (<= 65 536 points), one workgroup's sort
    (never for a source: it is below the plain limit), the library's radix sort with its digit histograms counted by the keys
    kernel into one of two sets used in turn -- and a context sees them in any order: every alignment must give the bits a
    fresh context gives."""
    from rsreg_amd import api
    tgt = rs.synth.render_frame(0, "N300", "bench")
    guess = rs.synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
    big_a, big_b = rs.synth.render_frame(1, "N300", "bench"), rs.synth.render_frame(2, "N300", "bench")
    mid = rs.PointCloud(big_a.points[::7].copy(), width=len(big_a.points[::7]), height=1, is_dense=False)     # 43 886: plain
    small = rs.PointCloud(big_b.points[::100].copy(), width=len(big_b.points[::100]), height=1, is_dense=False)
    edge = rs.PointCloud(big_a.points[:70000].copy(), width=70000, height=1, is_dense=False)                    # just above the plain limit
    order = [big_a, mid, big_b, small, edge, big_a, edge, big_b, big_b]

    def run(icp, src):
        icp.setInputSource(src)
        out = icp.align(guess)
        r = icp.result
        return (bytes(r.transform), bytes(r.sums_last), r.n_correspondences, r.iterations, np.stack([out.points[c] for c in "xyz"]).tobytes())

    def fresh():
        icp = api.IterativeClosestPoint(api.Context(0))
        icp.params = api.icp_params(max_iterations=3, criteria_mode=1, pipeline_mode=2, max_correspondence_distance=0.03)
        icp.setInputTarget(tgt)
        return icp

    one = fresh()
    for k, src in enumerate(order):
        assert run(one, src) == run(fresh(), src), k


@pytest.mark.parametrize("size", ["50k", "N300"])
def test_align_records_is_align_after_a_copy_of_the_input(api, rs, size):
    """rsreg_icp_align_records (PCL's `output = input` made on the way, incremental_icp.hpp:59) leaves byte for byte what
    rsreg_icp_align leaves in a copy of the source records: colours and padding the source's, xyz <- final * xyz, data[3] = 1.
    N300: more than one piece of 2^18 points comes home."""
    import ctypes as C
    from rsreg_amd import lib
    tgt, src = rs.synth.render_frame(0, size, "parity"), rs.synth.render_frame(1, size, "parity")
    recs = np.ascontiguousarray(src.points)
    raw = recs.view(np.uint8).reshape(len(recs), -1).copy()
    raw[:, 20:32] = np.arange(12, dtype=np.uint8) + 1          # the padding behind rgba travels too
    recs = raw.view(recs.dtype).reshape(-1)

    def clone(a):   # (numpy copies a structured array field by field: the padding would not travel)
        return np.frombuffer(bytearray(a.tobytes()), a.dtype)

    ctx = api.Context(0)
    L, prm = lib.lib(), api.icp_params(reference=True)
    outs = []
    for with_records in (False, True):
        res = lib.IcpResult()
        lib.check(L.rsreg_icp_set_source(ctx.h, recs.ctypes.data, len(recs), recs.dtype.itemsize, 0), ctx.h)
        lib.check(L.rsreg_icp_set_target(ctx.h, tgt.points.ctypes.data, len(tgt.points), tgt.points.dtype.itemsize, 0, prm.max_correspondence_distance), ctx.h)
        if with_records:
            out = np.full(len(recs), 0, recs.dtype)
            lib.check(L.rsreg_icp_align_records(ctx.h, None, C.byref(prm), C.byref(res), recs.ctypes.data, out.ctypes.data, out.dtype.itemsize), ctx.h)
        else:
            out = clone(recs)
            lib.check(L.rsreg_icp_align(ctx.h, None, C.byref(prm), C.byref(res), out.ctypes.data, out.dtype.itemsize), ctx.h)
        outs.append((out, np.array(res.transform)))
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    assert outs[0][0].tobytes() == outs[1][0].tobytes()
    assert not np.array_equal(outs[1][0]["x"], recs["x"])      # (something moved)
    # in place: aligned_out == source_records
    res = lib.IcpResult()
    inplace = clone(recs)
    lib.check(L.rsreg_icp_set_source(ctx.h, inplace.ctypes.data, len(recs), recs.dtype.itemsize, 0), ctx.h)
    lib.check(L.rsreg_icp_set_target(ctx.h, tgt.points.ctypes.data, len(tgt.points), tgt.points.dtype.itemsize, 0, prm.max_correspondence_distance), ctx.h)
    lib.check(L.rsreg_icp_align_records(ctx.h, None, C.byref(prm), C.byref(res), inplace.ctypes.data, inplace.ctypes.data, inplace.dtype.itemsize), ctx.h)
    assert inplace.tobytes() == outs[0][0].tobytes()
    # the Python and C++ adaptors go through it: the aligned cloud keeps the source's colours
    icp = _ref_icp(api, rs.PointCloud(clone(recs)), tgt)
    aligned = icp.align()
    assert aligned.points.tobytes() == outs[0][0].tobytes()
