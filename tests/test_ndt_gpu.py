"""GPU parity tests of the NDT path (voxel statistics kernel set + derivative pass + host
Newton / More-Thuente) through the C ABI, against the numpy golden vectors and the CPU oracle.
f64 statistics to 1e-10 relative; score / gradient / Hessian to 1e-9 relative of their scale;
final transform within 1e-4 Frobenius of the oracle (north-star tolerance)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(rs):
    from rsreg_amd import api as a, lib
    lib.build()
    if a.device_count() < 1:
        pytest.fail("no HIP device: the product has no CPU fallback")
    return a


def _ndt(api, src, tgt, **kw):
    n = api.NormalDistributionsTransform()
    n.params = api.ndt_params(reference=True, **kw)
    n.setInputSource(src)
    n.setInputTarget(tgt)
    return n


def test_voxel_statistics_match_golden_and_oracle(api, orc, golden):
    g = golden("ndt_small")
    tgt = np.ascontiguousarray(g["tgt"])
    n = _ndt(api, np.ascontiguousarray(g["src"]), tgt)
    m, c = n.voxels()
    np.testing.assert_array_equal(c, g["vox_n"])
    np.testing.assert_allclose(m[:, 0:3], g["vox_mean"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(m[:, 3:12].reshape(-1, 3, 3), g["vox_cov"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(m[:, 12:21].reshape(-1, 3, 3), g["vox_icov"], rtol=1e-7, atol=1e-9)
    o = orc.NdtOracle()
    o.set_target(tgt, 1.0)
    mo, co = o.voxels()
    np.testing.assert_array_equal(c, co)
    np.testing.assert_allclose(m, mo, rtol=1e-9, atol=1e-11)


def test_derivative_pass_matches_oracle_and_finite_differences(api, orc, golden):
    g = golden("ndt_small")
    tgt, src = np.ascontiguousarray(g["tgt"]), np.ascontiguousarray(g["src"])
    n = _ndt(api, src, tgt)
    score, grad, hess = n.derivatives(g["pose"])
    o = orc.NdtOracle()
    o.set_centroid_mode(1)   # the HIP path stores round(mean_f64) as the search centroid
    o.set_target(tgt, 1.0)
    so, go, ho = o.derivatives(src, g["pose"], orc.NdtParams.reference())
    assert abs(score - so) < 1e-10 * abs(so)
    np.testing.assert_allclose(grad, go, rtol=1e-9, atol=1e-9 * np.abs(go).max())
    np.testing.assert_allclose(hess, ho, rtol=1e-9, atol=1e-9 * np.abs(ho).max())
    # independent check: numpy score and finite-difference derivatives (float64 positions)
    assert abs(score - g["score"][0]) < 1e-6 * abs(g["score"][0])
    np.testing.assert_allclose(grad, g["grad_fd"], rtol=5e-4, atol=0.5)
    np.testing.assert_allclose(hess, g["hess_fd"], rtol=2e-3, atol=np.abs(g["hess_fd"]).max() * 2e-4)


@pytest.fixture(scope="module")
def edge_like(rs):
    """~30 k-point subsets standing in for the edge clouds of BASELINE configs[2]."""
    s = rs.synth
    out = []
    for k in (0, 2):
        f = s.render_frame(k, "N300", "bench").crop(0, 0, 640, 480, step=3)
        out.append(rs.PointCloud(np.ascontiguousarray(f.points[f.points["z"] != 0])))
    return out


@pytest.mark.parametrize("guess_kind", ["identity", "yaw"])
def test_align_matches_oracle(api, orc, rs, edge_like, guess_kind):
    tgt, src = edge_like
    guess = None if guess_kind == "identity" else rs.synth.small_transform(2.0, (0.0, 0.0, 0.0)).astype(np.float32)
    n = _ndt(api, src, tgt)
    out = n.align(guess)
    o = orc.NdtOracle()
    o.set_centroid_mode(1)
    o.set_target(tgt.points, 1.0)
    ro, oal = o.align(src.points, guess, orc.NdtParams.reference(), want_aligned=True)
    r = n.result
    assert (r.converged, r.iterations) == (ro.converged, ro.iterations)
    assert r.n_voxels == ro.n_voxels and r.n_derivative_passes == ro.n_derivative_passes
    err = np.linalg.norm(n.getFinalTransformation() - ro.T)
    assert err < 1e-5, err
    assert abs(r.score - ro.score) < 1e-7 * abs(ro.score)
    np.testing.assert_allclose(out.xyz, oal[:, :3], atol=2e-5)
    # PCL-faithful f32 running-sum centroids: still inside the north-star tolerance
    o2 = orc.NdtOracle()
    o2.set_target(tgt.points, 1.0)
    r2 = o2.align(src.points, guess, orc.NdtParams.reference())
    assert np.linalg.norm(n.getFinalTransformation() - r2.T) < 1e-4
    if guess is None:  # from a cold start the 1 m NDT must move towards the true pose
        gt = rs.synth.ground_truth(2, 0, "bench")
        assert np.linalg.norm(n.getFinalTransformation() - gt) < np.linalg.norm(np.eye(4) - gt)


@pytest.mark.parametrize("resolution", [1.0, 0.3])
def test_pcl_centroid_mode_is_bit_equal_to_the_oracles_pcl_mode(api, orc, rs, edge_like, resolution):
    """rsreg_ndt_set_centroid_mode(1): the voxels are searched by PCL's own centroid -- a float running sum over the
    voxel's points in input order, divided by float(n) (VoxelGridCovariance::applyFilter) -- bit for bit what the
    oracle's mode 0 computes, so that a pin against a real PCL build can be exact instead of "< 1e-4"; the alignment in
    that mode agrees with the oracle in that mode as tightly as the default modes agree with each other."""
    tgt, src = edge_like
    shuffled = rs.PointCloud(np.ascontiguousarray(tgt.points[np.random.default_rng(5).permutation(len(tgt))]))
    for cloud in (tgt, shuffled):                          # the sum depends on the input order: both orders must agree with the oracle
        n = api.NormalDistributionsTransform(api.default_context())
        n.params = api.ndt_params(reference=True)
        n.setResolution(resolution)
        n.setPclCentroids(True)
        n.setInputSource(src)
        n.setInputTarget(cloud)
        guess = rs.synth.small_transform(0.5, (0.0, 0.0, 0.0)).astype(np.float32)
        n.align(guess)
        o = orc.NdtOracle()
        o.set_centroid_mode(0)
        o.set_target(cloud.points, resolution)
        cg, co = n.centroids(), o.centroids()
        assert cg.shape == co.shape and len(cg) > 3
        np.testing.assert_array_equal(cg.view(np.uint32), co.view(np.uint32))
        po = orc.NdtParams.reference()
        po.resolution = resolution
        ro = o.align(src.points, guess, po)
        r = n.result
        assert (r.converged, r.iterations, r.n_voxels, r.n_derivative_passes) == (ro.converged, ro.iterations, ro.n_voxels, ro.n_derivative_passes)
        assert np.linalg.norm(n.getFinalTransformation() - ro.T) < 1e-5
    # the default mode is what it was: the rounded f64 mean
    d = api.NormalDistributionsTransform(api.default_context())
    d.params = api.ndt_params(reference=True)
    d.setResolution(resolution)
    d.setInputSource(src)
    d.setInputTarget(tgt)
    d.align()
    m, _ = d.voxels()
    np.testing.assert_array_equal(d.centroids(), m[:, 0:3].astype(np.float32))


@pytest.mark.parametrize("resolution", [0.3, 0.12])
def test_many_voxels_several_table_chunks(api, orc, rs, edge_like, resolution):
    """The derivative pass stages the voxel table through LDS 64 voxels at a time and deals (point, voxel) pairs to the
    lanes chunk by chunk: a fine grid (hundreds to thousands of voxels, ragged last chunk) against the oracle, for the
    three kinds of pass (score + gradient + Hessian in derivatives(), all of them inside align())."""
    tgt, src = edge_like
    n = _ndt(api, src, tgt, resolution=resolution)
    o = orc.NdtOracle()
    o.set_centroid_mode(1)
    o.set_target(tgt.points, resolution)
    po = orc.NdtParams.reference()
    po.resolution = resolution
    pose = np.array([0.01, -0.02, 0.015, 0.004, -0.006, 0.003])
    score, grad, hess = n.derivatives(pose)
    so, go, ho = o.derivatives(src.points, pose, po)
    m, c = n.voxels()
    assert len(c) > 2 * 64 and len(c) % 64 != 0, len(c)
    assert abs(score - so) < 1e-10 * abs(so)
    np.testing.assert_allclose(grad, go, rtol=1e-9, atol=1e-9 * np.abs(go).max())
    np.testing.assert_allclose(hess, ho, rtol=1e-9, atol=1e-9 * np.abs(ho).max())
    guess = rs.synth.small_transform(0.5, (0.0, 0.0, 0.0)).astype(np.float32)
    n.align(guess)
    ro = o.align(src.points, guess, po)
    r = n.result
    assert (r.converged, r.iterations, r.n_voxels, r.n_derivative_passes) == (ro.converged, ro.iterations, ro.n_voxels, ro.n_derivative_passes)
    assert np.linalg.norm(n.getFinalTransformation() - ro.T) < 1e-5


def test_ndt_then_icp_pair_like_the_reference_scheme(api, orc, rs, edge_like):
    """configs[2] shape: NDT on the subset gives the guess, ICP refines (ndt_edge...hpp:71-99)."""
    tgt, src = edge_like
    n = _ndt(api, src, tgt)
    aligned = n.align(rs.synth.small_transform(1.0, (0, 0, 0)).astype(np.float32))
    icp = api.IterativeClosestPoint()
    icp.params = api.icp_params(reference=True)
    icp.setInputSource(aligned)
    icp.setInputTarget(tgt)
    icp.align()
    o = orc.IcpOracle()
    o.set_target(tgt.points)
    o.set_source(aligned.points)
    ro = o.align(None, orc.IcpParams.reference())
    assert icp.hasConverged() == bool(ro.converged)
    np.testing.assert_allclose(icp.getFinalTransformation(), ro.T, atol=2e-6)


def test_ndt_edge_cases(api, rs):
    pts = np.zeros((100, 4), np.float32)
    pts[:, 0] = np.linspace(0, 0.5, 100)
    n = _ndt(api, pts, pts[:4])            # fewer than 6 points per voxel: no voxels
    n.align()
    assert n.result.n_voxels == 0 and n.result.iterations == 0
    np.testing.assert_array_equal(n.getFinalTransformation(), np.eye(4, dtype=np.float32))
    bad = pts.copy()
    bad[3, 1] = np.nan
    n = _ndt(api, bad, pts)                # collinear target: eigenvalue floor keeps it usable
    n.align()
    assert np.isfinite(n.getFinalTransformation()).all()


def test_line_search_in_one_launch_gives_the_same_bits(api, rs, monkeypatch):
    """RSREG_NDT_RESIDENT_LS=1: all the derivative passes of a More-Thuente line search in one launch (k_ndt_line_search: 512
    resident workgroups, grid-wide counts, the state machine of csrc/ndt_math.hpp advanced on the device) against the default,
    a launch pair and a wait per pass (ndt_edge_based_registration.hpp:38-43,86-92).  Same sums in the same order through the
    same source: the same transform, score and pass count, bit for bit."""
    tgt, src = rs.synth.render_frame(0, "50k", "parity"), rs.synth.render_frame(1, "50k", "parity")
    guess = rs.synth.small_transform(0.4, (0.01, -0.005, 0.008)).astype(np.float32)

    def run():
        n = api.NormalDistributionsTransform(api.Context(0))   # (a context looks at the environment when it is created)
        n.params = api.ndt_params(reference=True)
        n.setInputSource(src)
        n.setInputTarget(tgt)
        out = n.align(guess)
        r = n.result
        return (bytes(r.transform), r.score, r.iterations, r.n_derivative_passes, r.converged, np.stack([out.points[k] for k in "xyz"]).tobytes())

    monkeypatch.delenv("RSREG_NDT_RESIDENT_LS", raising=False)
    base = run()
    monkeypatch.setenv("RSREG_NDT_RESIDENT_LS", "1")
    for _ in range(3):
        assert run() == base
    monkeypatch.delenv("RSREG_NDT_RESIDENT_LS")
    assert base[3] > 3   # several passes: a line search did run


def test_pass_with_its_reduce_in_one_launch_gives_the_same_bits(api, rs, monkeypatch):
    """Round 6, RSREG_NDT_ONE_LAUNCH=1: a derivative pass as ONE launch whose last workgroup adds the 512 slabs (k_ndt_pass_reduce;
    opt-in: measured slower) -- against the default launch pair (k_ndt_pass + k_ndt_final_reduce): the same summation tree, so the same
    28 sums of every pass and the same transform, score and pass count, bit for bit; the derivatives entry point returns the same doubles."""
    tgt, src = rs.synth.render_frame(0, "50k", "parity"), rs.synth.render_frame(1, "50k", "parity")
    guess = rs.synth.small_transform(0.4, (0.01, -0.005, 0.008)).astype(np.float32)

    def run():
        n = api.NormalDistributionsTransform(api.Context(0))   # (a context looks at the environment when it is created)
        n.params = api.ndt_params(reference=True)
        n.setInputSource(src)
        n.setInputTarget(tgt)
        out = n.align(guess)
        r = n.result
        score, grad, hess = n.derivatives(np.array([0.01, -0.005, 0.008, 0.002, -0.004, 0.007]))
        return (bytes(r.transform), r.score, r.iterations, r.n_derivative_passes, r.converged, np.stack([out.points[k] for k in "xyz"]).tobytes(),
                float(score), np.asarray(grad).tobytes(), np.asarray(hess).tobytes())

    monkeypatch.delenv("RSREG_NDT_ONE_LAUNCH", raising=False)
    two = run()
    monkeypatch.setenv("RSREG_NDT_ONE_LAUNCH", "1")
    for _ in range(3):
        assert run() == two
    monkeypatch.delenv("RSREG_NDT_ONE_LAUNCH")
    assert two[3] > 3


def test_target_grid_from_a_known_box_is_the_grid_from_the_measured_one(api, rs, edge_like, monkeypatch):
    """A cloud handle that knows a box AROUND its points (the union of a measured box and the transformed corners of another:
    what the NDT-edge loop's grown target carries) builds its voxel grid without measuring the box: the leaves are
    floor(x / leaf) whatever the grid's origin, and they are visited in (z, y, x) order whatever its extent -- the same voxels
    in the same order, the same alignment bit for bit, as from the box measured on the records (RSREG_NO_BOX_CACHE=1)."""
    src, tgt = edge_like
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = (0.03, -0.02, 0.04)
    c, s = np.cos(0.02), np.sin(0.02)
    T[0, 0], T[0, 2], T[2, 0], T[2, 2] = c, s, -s, c

    def run():
        ctx = api.Context(0)   # (a context looks at the environment when it is created)
        a, b = api.DeviceCloud(src, ctx=ctx), api.DeviceCloud(tgt, ctx=ctx)
        icp = api.IterativeClosestPoint(ctx)   # (an ICP index build measures b's box and a source load a's; the transform's output carries a's, moved)
        icp.setMaxCorrespondenceDistance(0.05)
        icp.setMaximumIterations(1)
        icp.setInputSource(a)
        icp.setInputTarget(b)
        icp.align()
        moved = api.transformPointCloud(a, T)
        grown = moved + b
        ndt = api.NormalDistributionsTransform(ctx)
        ndt.params = api.ndt_params(reference=True)
        ndt.setInputSource(a)
        ndt.setInputTarget(grown)
        ndt.align()
        m, cnt = ndt.voxels()
        r = ndt.result
        return m.tobytes(), cnt.tobytes(), bytes(r.transform), r.score, r.iterations, r.n_derivative_passes, len(cnt)

    monkeypatch.delenv("RSREG_NO_BOX_CACHE", raising=False)
    with_box = run()
    monkeypatch.setenv("RSREG_NO_BOX_CACHE", "1")
    measured = run()
    monkeypatch.delenv("RSREG_NO_BOX_CACHE")
    assert with_box[-1] > 3
    assert with_box == measured
