"""The optional correspondence filters of pcl::IterativeClosestPoint: reciprocal correspondences and
the trimmed rejector (what the reference evidently intended: it constructs a CorrespondenceRejectorTrimmed
and never attaches it, src/incremental_icp.hpp:38).  Off by default, so the reference path is untouched.
CPU: the oracle's filters against an independent brute-force numpy restatement.  GPU: the engine against
the oracle, correspondence for correspondence."""
import numpy as np
import pytest


def _np_filters(src, tgt, gate, reciprocal, ratio):
    """Brute force in float32 with FLANN's summation order; ties: lowest index."""
    def d2(a, b):
        dx, dy, dz = (a[:, None, k] - b[None, :, k] for k in range(3))
        return ((dx * dx + dy * dy).astype(np.float32) + dz * dz).astype(np.float32)
    D = d2(src, tgt)
    idx = D.argmin(1)
    dist = D[np.arange(len(src)), idx]
    ok = ~(dist.astype(np.float64) > gate * gate)
    if reciprocal:
        back = d2(tgt, src).argmin(1)                    # nearest source point of every target point
        ok &= back[idx] == np.arange(len(src))
    if 0 < ratio < 1:
        cand = np.nonzero(ok)[0]
        keep = int(np.floor(np.float32(ratio) * np.float32(len(cand))))
        if keep < len(cand):
            order = cand[np.lexsort((cand, dist[cand]))]
            ok[order[keep:]] = False
    return np.where(ok, idx, -1), dist


@pytest.mark.parametrize("reciprocal,ratio", [(1, 0.0), (0, 0.6), (1, 0.45)])
def test_oracle_filters_match_brute_force(orc, reciprocal, ratio):
    rng = np.random.default_rng(5)
    tgt = rng.random((700, 3)).astype(np.float32)
    src = (tgt[rng.integers(0, 700, 900)] + rng.normal(0, 0.02, (900, 3))).astype(np.float32)
    src[100:110] = src[100]                              # duplicates: only the first copy can be reciprocal
    o = orc.IcpOracle()
    o.set_target(tgt)
    o.set_source(src)
    p = orc.IcpParams.default()
    p.max_correspondence_distance, p.use_reciprocal, p.trim_overlap_ratio = 0.05, reciprocal, ratio
    o.begin(None, p)
    oi, od = o.search()
    ni, nd = _np_filters(src, tgt, 0.05, reciprocal, ratio)
    np.testing.assert_array_equal(oi, ni)
    np.testing.assert_array_equal(od[ni >= 0], nd[ni >= 0])
    if reciprocal:
        assert (oi[101:110] == -1).all()
    if ratio:
        base = _np_filters(src, tgt, 0.05, reciprocal, 0.0)[0]
        assert (oi >= 0).sum() == int(np.floor(np.float32(ratio) * np.float32((base >= 0).sum())))


@pytest.fixture(scope="module")
def api(rs):
    from rsreg_amd import api as a, lib
    lib.build()
    if a.device_count() < 1:
        pytest.fail("no HIP device: the product has no CPU fallback")
    return a


@pytest.mark.gpu
@pytest.mark.parametrize("reciprocal,ratio", [(1, 0.0), (0, 0.5), (1, 0.7)])
@pytest.mark.parametrize("size,preset,gate", [("50k", "parity", 0.01), ("50k", "bench", 0.05)])
def test_gpu_filters_match_oracle(api, orc, rs, reciprocal, ratio, size, preset, gate):
    tgt, src = rs.synth.render_frame(0, size, preset), rs.synth.render_frame(1, size, preset)
    guess = None if preset == "parity" else rs.synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
    icp = api.IterativeClosestPoint()
    icp.params = api.icp_params(max_iterations=5, criteria_mode=1, max_correspondence_distance=gate)
    icp.setUseReciprocalCorrespondences(reciprocal)
    icp.setTrimmedRejectorOverlapRatio(ratio)
    icp.setInputSource(src)
    icp.setInputTarget(tgt)
    o = orc.IcpOracle()
    o.set_target(tgt.points, dedup=True)
    o.set_source(src.points)
    p = orc.IcpParams.default()
    p.max_iterations, p.criteria_mode, p.max_correspondence_distance, p.num_threads = 5, 1, gate, 8
    p.use_reciprocal, p.trim_overlap_ratio = reciprocal, ratio
    # first iteration stage by stage
    icp.begin(guess)
    o.begin(guess, p)
    gi, gd = icp.search()
    oi, od = o.search()
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd[oi >= 0], od[oi >= 0])
    gs, os_ = icp.sums(), o.sums()
    np.testing.assert_allclose(gs, os_, rtol=1e-11, atol=1e-11)
    assert gs[0] == os_[0] == (oi >= 0).sum() > 100
    icp.end()
    plain = api.IterativeClosestPoint(api.Context(0))     # (its own context: a context carries one alignment at a time)
    plain.params = api.icp_params(max_iterations=5, criteria_mode=1, max_correspondence_distance=gate)
    plain.setInputSource(src)
    plain.setInputTarget(tgt)
    plain.begin(guess)
    pi_, _ = plain.search()
    plain.end()
    assert (oi >= 0).sum() < (pi_ >= 0).sum()            # the filters do remove pairs
    if ratio and not reciprocal:
        assert (oi >= 0).sum() == int(np.floor(np.float32(ratio) * np.float32((pi_ >= 0).sum())))
    # whole alignments (whatever pipeline is asked for, the filters run between the staged kernels)
    for pipeline in (0, 2):
        icp.params.pipeline_mode = pipeline
        icp.align(guess)
        r = o.align(guess, p)
        assert (icp.result.iterations, icp.result.n_correspondences) == (r.iterations, r.n_correspondences)
        assert np.linalg.norm(icp.getFinalTransformation() - r.T) < 1e-5
