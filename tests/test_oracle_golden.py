"""CPU: the C oracle (oracle/*.c) against the independent numpy/scipy golden vectors
(tests/golden, made by oracle/make_golden.py).  Integer/index results bit-exact, float32
distances bit-exact, f64 sums to 1e-12 relative, transforms to 2e-6 absolute."""
import numpy as np
import pytest

T_TOL = 2e-6


def _check_search(g, prefix, it, idx, d2, tgt_xyz, cur_xyz):
    gi, gd = g["%s_it%d_index" % (prefix, it)], g["%s_it%d_sqr_dist" % (prefix, it)]
    acc = gi >= 0
    assert ((idx >= 0) == acc).all(), "gate decisions differ"
    np.testing.assert_array_equal(d2[acc], gd[acc])  # float32 bit-exact
    diff = acc & (idx != gi)
    if diff.any():  # allowed only for exactly equidistant targets (tie-break unspecified)
        a, b = tgt_xyz[idx[diff]], tgt_xyz[gi[diff]]
        q = cur_xyz[diff]
        da = ((q - a) ** 2)
        db = ((q - b) ** 2)
        da = (da[:, 0] + da[:, 1]) + da[:, 2]
        db = (db[:, 0] + db[:, 1]) + db[:, 2]
        np.testing.assert_array_equal(da, db)


@pytest.mark.parametrize("case,prefix", [("kat_exact", "ref"), ("crop_parity", "ref")])
def test_icp_reference_params_stagewise(orc, golden, case, prefix):
    g = golden(case)
    o = orc.IcpOracle()
    o.set_target(g["tgt"])
    o.set_source(g["src"])
    p = orc.IcpParams.reference()
    o.begin(g["guess"], p)
    idx, d2 = o.search()
    tgt = np.stack([g["tgt"]["x"], g["tgt"]["y"], g["tgt"]["z"]], 1)
    _check_search(g, prefix, 0, idx, d2, tgt, o.current())
    s = o.sums()
    np.testing.assert_allclose(s, g[prefix + "_it0_sums"], rtol=1e-12, atol=1e-12)
    t_inc, done = o.update()
    np.testing.assert_allclose(t_inc, g[prefix + "_it0_t_inc"], atol=T_TOL)
    r = o.end()
    it, state, conv = g[prefix + "_meta"]
    assert (r.iterations, r.state, r.converged) == (it, state, conv)
    assert done and r.iterations == 1  # the reference's epsilons stop after 1 iteration
    np.testing.assert_allclose(r.T, g[prefix + "_final"], atol=T_TOL)


def test_kat_recovers_transform(orc, golden):
    g = golden("kat_exact")
    o = orc.IcpOracle()
    o.set_target(g["tgt"])
    o.set_source(g["src"])
    for accum in (0, 1):
        p = orc.IcpParams.reference()
        p.accum_mode = accum
        r, aligned = o.align(None, p, want_aligned=True)
        assert np.abs(r.T - g["T_true"]).max() < (2e-5 if accum == 0 else 2e-6)
        tgt = np.stack([g["tgt"]["x"], g["tgt"]["y"], g["tgt"]["z"]], 1)
        assert np.abs(aligned[:, :3] - tgt).max() < 1e-4
        assert (aligned[:, 3] == 1).all()


@pytest.mark.parametrize("nn_mode", [0, 1])
@pytest.mark.parametrize("dedup", [0, 1])
def test_icp_fixed_iterations(orc, golden, nn_mode, dedup):
    g = golden("crop_parity")
    o = orc.IcpOracle()
    o.set_target(g["tgt"], dedup=bool(dedup))
    o.set_source(g["src"])
    p = orc.IcpParams.default()
    p.max_iterations, p.criteria_mode, p.nn_mode = 8, 1, nn_mode
    p.max_correspondence_distance = 0.02
    p.transformation_epsilon = 1e-12
    p.euclidean_fitness_epsilon = 1e-12
    r = o.align(g["guess"], p)
    assert (r.iterations, r.state, r.converged) == tuple(g["fixed8_meta"])
    np.testing.assert_allclose(r.T, g["fixed8_final"], atol=5e-6)
    np.testing.assert_allclose(r.mse, g["fixed8_mse"][0], rtol=1e-4)


def test_icp_pcl_criteria_with_guess(orc, golden):
    g = golden("crop_bench")
    o = orc.IcpOracle()
    o.set_target(g["tgt"])
    o.set_source(g["src"])
    p = orc.IcpParams.default()
    p.max_iterations = 30
    p.max_correspondence_distance = 0.05
    p.transformation_epsilon = 1e-9
    p.euclidean_fitness_epsilon = 1e-7
    o.begin(g["guess"], p)
    idx, d2 = o.search()
    tgt = np.stack([g["tgt"]["x"], g["tgt"]["y"], g["tgt"]["z"]], 1)
    _check_search(g, "pcl", 0, idx, d2, tgt, o.current())
    r = o.align(g["guess"], p)
    assert (r.iterations, r.state, r.converged) == tuple(g["pcl_meta"])
    np.testing.assert_allclose(r.T, g["pcl_final"], atol=2e-5)


def test_f32_accumulation_mode_within_tolerance(orc, golden):
    """PCL sums in float; the f64 path must stay within the 1e-4 Frobenius bar of it."""
    g = golden("crop_parity")
    o = orc.IcpOracle()
    o.set_target(g["tgt"])
    o.set_source(g["src"])
    Ts = []
    for accum in (0, 1):
        p = orc.IcpParams.reference()
        p.accum_mode = accum
        Ts.append(o.align(None, p).T)
    assert np.linalg.norm(Ts[0] - Ts[1]) < 1e-4


def test_no_correspondences(orc):
    src = np.zeros((10, 4), np.float32)
    tgt = np.ones((10, 4), np.float32) * 5
    tgt[:, 0] += np.arange(10)
    o = orc.IcpOracle()
    o.set_target(tgt)
    o.set_source(src)
    r = o.align(None, orc.IcpParams.reference())
    assert r.converged == 0 and r.state == 5 and r.iterations == 0
    np.testing.assert_array_equal(r.T, np.eye(4, dtype=np.float32))


def test_nonfinite_points_are_skipped(orc, golden):
    g = golden("kat_exact")
    src, tgt = g["src"].copy(), g["tgt"].copy()
    src["x"][5] = np.nan
    tgt["y"][7] = np.inf
    o = orc.IcpOracle()
    o.set_target(tgt)
    o.set_source(src)
    p = orc.IcpParams.reference()
    o.begin(None, p)
    idx, d2 = o.search()
    assert idx[5] == -1 and idx[7] == -1 and (idx[[4, 6, 8]] == [4, 6, 8]).all()
    r = o.align(None, p)
    assert np.abs(r.T - g["T_true"]).max() < 2e-6


def test_approx_voxel_grid(orc, golden):
    g = golden("approx_voxel")
    for key_in, key_out, leaf in (("in", "leaf_001", 0.01), ("in", "leaf_1", 1.0), ("wide_in", "wide_leaf_01", 0.1)):
        out = orc.approx_voxel_grid(g[key_in], (leaf, leaf, leaf))
        exp = g[key_out]
        assert len(out) == len(exp)
        for f in ("x", "y", "z", "w", "rgba"):
            np.testing.assert_array_equal(out[f], exp[f])


def test_transform_cloud(orc, golden):
    g = golden("crop_parity")
    T = g["ref_final"]
    pts = g["src"].copy()
    pts["x"][3] = np.nan
    out = orc.transform_cloud(pts, T, is_dense=False)
    x = np.stack([pts["x"], pts["y"], pts["z"]], 1)
    exp = np.empty_like(x)
    for r in range(3):
        exp[:, r] = ((T[r, 0] * x[:, 0] + T[r, 1] * x[:, 1]) + T[r, 2] * x[:, 2]) + T[r, 3]
    ok = np.isfinite(x).all(1)
    got = np.stack([out["x"], out["y"], out["z"]], 1)
    np.testing.assert_array_equal(got[ok], exp[ok])
    assert np.isnan(got[3, 0]) and got[3, 1] == pts["y"][3]
    np.testing.assert_array_equal(out["rgba"], pts["rgba"])


def test_ndt_voxels_and_derivatives(orc, golden):
    g = golden("ndt_small")
    n = orc.NdtOracle()
    n.set_target(np.ascontiguousarray(g["tgt"]), 1.0)
    m, c = n.voxels()
    np.testing.assert_array_equal(c, g["vox_n"])
    np.testing.assert_allclose(m[:, 0:3], g["vox_mean"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(m[:, 3:12].reshape(-1, 3, 3), g["vox_cov"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(m[:, 12:21].reshape(-1, 3, 3), g["vox_icov"], rtol=1e-7, atol=1e-9)
    p = orc.NdtParams.reference()
    score, grad, hess = n.derivatives(np.ascontiguousarray(g["src"]), g["pose"], p)
    assert abs(score - g["score"][0]) < 1e-6 * abs(g["score"][0])
    # PCL evaluates the derivatives at the float32-transformed cloud; the finite differences
    # of the golden use float64 positions, so agreement is limited to ~1e-4 relative.
    np.testing.assert_allclose(grad, g["grad_fd"], rtol=5e-4, atol=0.5)
    np.testing.assert_allclose(hess, g["hess_fd"], rtol=2e-3, atol=np.abs(g["hess_fd"]).max() * 2e-4)
    np.testing.assert_allclose(hess, hess.T, rtol=1e-9, atol=1e-6)


def test_ndt_align_improves_pose(orc, rs):
    synth = rs.synth
    tgt = synth.render_frame(0, "N300", "bench").crop(0, 0, 640, 480, step=6)
    src = synth.render_frame(2, "N300", "bench").crop(0, 0, 640, 480, step=6)
    tx = tgt.xyz[tgt.points["z"] != 0]
    sx = src.xyz[src.points["z"] != 0]
    n = orc.NdtOracle()
    n.set_target(np.ascontiguousarray(tx), 1.0)
    p = orc.NdtParams.reference()
    r = n.align(np.ascontiguousarray(sx), None, p)
    gt = synth.ground_truth(2, 0, "bench")
    err0 = np.linalg.norm(np.eye(4) - gt)
    err = np.linalg.norm(r.T - gt)
    assert r.converged and r.iterations >= 1
    assert err < 0.5 * err0, (err, err0)
