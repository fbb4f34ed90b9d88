"""GPU: the one-launch "flag, scan, scatter" of the index build (csrc/compact.hpp, k_dense_compact) and the source load's
copies-to-weights step (k_source_unique, the weights formed by the first search launch) on clouds large enough to take those paths (the library's own radix sort from
65 536 points on; smaller clouds go through rocPRIM's scan and the caller's order), with what a look-back and a run search
get wrong first: runs of exact copies of every length around the 256-thread and 4 096-record boundaries, a record count
that is no multiple of anything, non-finite records in between.

The checker shares nothing with the engine: scipy's k-d tree over the DISTINCT target points proposes 8 candidates per
query, their distances are recomputed in float32 in FLANN's order and the lowest (distance, lowest original index) wins
(the tie-break among equidistant points is the engine's and the oracle's convention)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
scipy_spatial = pytest.importorskip("scipy.spatial")


@pytest.fixture(scope="module")
def api(rs):
    from rsreg_amd import api as a, lib
    lib.build()
    if a.device_count() < 1:
        pytest.fail("no HIP device: the product has no CPU fallback")
    return a


RUNS = [1, 2, 3, 5, 63, 64, 65, 255, 256, 257, 1000, 4095, 4096, 4097, 9001]


def room(rng, n):
    """points on three walls of a 2 m room, a few mm of noise: the density the dense table is built for"""
    p = rng.uniform(0.0, 2.0, (n, 3))
    w = rng.integers(0, 3, n)
    p[np.arange(n), w] = 0.002 * rng.standard_normal(n)
    return p.astype(np.float32)


def with_copies(rng, base, runs, n_bad):
    """base points + `run` exact copies of a randomly chosen base point for every run, shuffled, + non-finite records"""
    extra = [np.repeat(base[rng.integers(0, len(base))][None, :], r, axis=0) for r in runs]
    bad = np.full((n_bad, 3), np.nan, np.float32)
    bad[::2] = np.inf
    pts = np.concatenate([base] + extra + [bad]).astype(np.float32)
    return pts[rng.permutation(len(pts))]


def nearest(src, tgt, gate):
    """(index, float32 squared distance) of every source point's nearest target point, -1 beyond the gate or for a
    non-finite query; lowest original index among equidistant points"""
    ok_t = np.isfinite(tgt).all(1)
    ti = np.nonzero(ok_t)[0]
    uniq, first = np.unique(tgt[ok_t], axis=0, return_index=True)
    first = ti[first]                                   # lowest original index of every distinct point (np.unique: first occurrence)
    ok_s = np.isfinite(src).all(1)
    k = min(8, len(uniq))
    _, cand = scipy_spatial.cKDTree(uniq.astype(np.float64)).query(src[ok_s].astype(np.float64), k=k)
    cand = cand.reshape(len(cand), -1)
    q = src[ok_s][:, None, :]
    c = uniq[cand]
    dx, dy, dz = q[..., 0] - c[..., 0], q[..., 1] - c[..., 1], q[..., 2] - c[..., 2]
    d2 = (dx * dx + dy * dy) + dz * dz                  # float32, FLANN's L2_Simple order
    bd2 = d2.min(axis=1)
    bidx = np.where(d2 == bd2[:, None], first[cand], np.iinfo(np.int64).max).min(axis=1)   # lowest index among the equidistant
    idx = np.full(len(src), -1, np.int64)
    out_d2 = np.zeros(len(src), np.float32)
    inside = ~(bd2.astype(np.float64) > float(gate) * float(gate))
    sel = np.nonzero(ok_s)[0]
    idx[sel[inside]] = bidx[inside]
    out_d2[sel[inside]] = bd2[inside]
    return idx, out_d2


@pytest.mark.parametrize("seed", [0, 1])
def test_index_build_with_copies_across_workgroup_boundaries(api, rs, seed):
    rng = np.random.default_rng(900 + seed)
    base = room(rng, 70001 + 977 * seed)
    tgt = with_copies(rng, base, RUNS + RUNS[::-1], 37)
    src = (base[rng.permutation(len(base))[:3000]] + rng.normal(0, 0.003, (3000, 3))).astype(np.float32)
    gate = 0.05
    icp = api.IterativeClosestPoint(api.Context(0))
    icp.params = api.icp_params(max_iterations=2, criteria_mode=1, pipeline_mode=0, max_correspondence_distance=gate)
    icp.setInputSource(rs.PointCloud.from_xyz(src))
    icp.setInputTarget(rs.PointCloud.from_xyz(tgt))
    icp.begin()
    gi = icp.grid_info()
    assert gi.index_kind == 1                                         # dense table: the path under test
    n_finite = int(np.isfinite(tgt).all(1).sum())
    n_distinct = len(np.unique(tgt[np.isfinite(tgt).all(1)], axis=0))
    # copies are dropped where they follow each other in the sorted order (same cell, same x bucket): never more records
    # than finite points, never fewer than distinct ones
    assert n_distinct <= gi.n_unique_points <= n_finite
    want_idx, want_d2 = nearest(src, tgt, gate)
    for _ in range(2):                                                # the second round starts from the first one's seeds
        idx, d2 = icp.search()
        assert np.array_equal(idx.astype(np.int64), want_idx)
        assert np.array_equal(d2[want_idx >= 0], want_d2[want_idx >= 0])
    icp.end()


@pytest.mark.parametrize("pipeline", [1, 2])
def test_source_copies_become_weights(api, rs, pipeline):
    """A source of > 65 536 points is put into spatial order and its exact copies are merged into weighted points
    (k_source_unique): every original point must still get its correspondence, and the 17 sums must be those of all the
    original points, copies counted as often as they occur."""
    rng = np.random.default_rng(41)
    tgt = room(rng, 90000)
    base = (tgt[rng.permutation(len(tgt))[:66000]] + rng.normal(0, 0.002, (66000, 3))).astype(np.float32)
    src = with_copies(rng, base, RUNS, 21)
    gate = 0.02
    icp = api.IterativeClosestPoint(api.Context(0))
    icp.params = api.icp_params(max_iterations=1, criteria_mode=1, pipeline_mode=0, max_correspondence_distance=gate)
    icp.setInputSource(rs.PointCloud.from_xyz(src))
    icp.setInputTarget(rs.PointCloud.from_xyz(tgt))
    icp.begin()
    gi = icp.grid_info()
    n_distinct = len(np.unique(src[np.isfinite(src).all(1)], axis=0)) + int((~np.isfinite(src).all(1)).sum())
    assert n_distinct <= gi.n_source_distinct < len(src)              # copies were merged (non-finite records never are)
    want_idx, want_d2 = nearest(src, tgt, gate)
    idx, d2 = icp.search()
    assert np.array_equal(idx.astype(np.int64), want_idx)             # per ORIGINAL source point, copies included
    assert np.array_equal(d2[want_idx >= 0], want_d2[want_idx >= 0])
    sums = icp.sums()
    icp.end()
    m = want_idx >= 0
    P, Q = src[m].astype(np.float64), tgt[want_idx[m]].astype(np.float64)
    want = np.concatenate([[m.sum()], P.sum(0), Q.sum(0), (Q[:, :, None] * P[:, None, :]).sum(0).ravel(), [want_d2[m].astype(np.float64).sum()]])
    assert sums[0] == want[0]                                         # an exact integer: every copy counted
    assert np.allclose(sums, want, rtol=1e-11, atol=1e-9)
    # and a whole alignment of the fused pipelines (weights inside the search launch, restart inside its first launch)
    fused = api.IterativeClosestPoint(api.Context(0))
    fused.params = api.icp_params(max_iterations=1, criteria_mode=1, pipeline_mode=pipeline, max_correspondence_distance=gate)
    fused.setInputSource(rs.PointCloud.from_xyz(src))
    fused.setInputTarget(rs.PointCloud.from_xyz(tgt))
    fused.align()
    got = np.array(fused.result.sums_last)
    assert got[0] == want[0]
    assert np.allclose(got, want, rtol=1e-11, atol=1e-9)
