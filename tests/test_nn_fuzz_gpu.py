"""GPU: randomized exactness test of the correspondence search.  Many small random scenes --
clustered, planar, lattice (many exact distance ties), duplicated points, queries far outside
the target's box, gates from a fraction of a cell to unbounded -- each checked against a float32
brute force that follows FLANN's L2_Simple order ((dx^2 + dy^2) + dz^2) with the canonical
tie-break (lowest target index), and PCL's gate (reject iff d^2 > max_dist^2 in double).
Two search rounds per scene: the second starts from the seeds the first one left."""
import os

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


def brute(src, tgt, gate):
    s = src.astype(np.float32)
    t = tgt.astype(np.float32)
    idx = np.full(len(s), -1, np.int64)
    d2o = np.zeros(len(s), np.float32)
    ok_t = np.isfinite(t).all(1)
    ti = np.nonzero(ok_t)[0]
    tt = t[ok_t]
    gate2 = float(gate) * float(gate)
    for i in range(len(s)):
        if not np.isfinite(s[i]).all() or len(tt) == 0:
            continue
        d = s[i] - tt
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]   # float32, FLANN's order
        j = int(np.argmin(d2))                                             # first minimum = lowest index
        if not (float(d2[j]) > gate2):
            idx[i] = ti[j]
            d2o[i] = d2[j]
    return idx, d2o


def scene(rng, kind, n):
    if kind == "uniform":
        return rng.uniform(-0.3, 0.3, (n, 3))
    if kind == "plane":
        p = rng.uniform(-0.4, 0.4, (n, 3))
        p[:, 2] = 1.0 + 0.002 * rng.standard_normal(n)
        return p
    if kind == "clusters":
        c = rng.uniform(-0.5, 0.5, (6, 3))
        return c[rng.integers(0, 6, n)] + 0.01 * rng.standard_normal((n, 3)) * rng.uniform(0.1, 3.0)
    if kind == "lattice":       # exact ties everywhere
        g = rng.integers(0, 12, (n, 3)).astype(np.float64)
        return g * 0.0078125
    if kind == "line":
        t = rng.uniform(0, 1, n)
        return np.stack([t, 0.5 * t, np.full(n, 0.25)], 1)
    raise ValueError(kind)


@pytest.mark.parametrize("seed", range(6))
def test_search_matches_brute_force_on_random_scenes(api, rs, seed):
    rng = np.random.default_rng(1234 + seed)
    kinds = ["uniform", "plane", "clusters", "lattice", "line"]
    gates = [0.004, 0.013, 0.05, 0.2, 1e30]
    n_checked = 0
    for rep in range(16):
        kind = kinds[(seed + rep) % len(kinds)]
        nt, ns = int(rng.integers(1, 3000)), int(rng.integers(1, 1500))
        tgt = scene(rng, kind, nt).astype(np.float32)
        src = (scene(rng, kind, ns) + rng.uniform(-0.02, 0.02, 3)).astype(np.float32)
        if rep % 3 == 0:       # duplicates (the RealSense (0,0,0) convention) and invalid records
            tgt[rng.integers(0, nt, max(1, nt // 10))] = 0.0
            src[rng.integers(0, ns, max(1, ns // 10))] = 0.0
            tgt[rng.integers(0, nt, max(1, nt // 50))] = np.nan
            src[rng.integers(0, ns, max(1, ns // 50))] = np.inf
        if rep % 4 == 1:       # some queries far outside the target's box
            src[rng.integers(0, ns, max(1, ns // 8))] += rng.uniform(-3, 3, 3).astype(np.float32)
        gate = gates[int(rng.integers(0, len(gates)))]
        tc, sc = rs.PointCloud.from_xyz(tgt), rs.PointCloud.from_xyz(src)
        icp = api.IterativeClosestPoint()
        icp.params = api.icp_params(max_iterations=3, criteria_mode=1, pipeline_mode=0, max_correspondence_distance=gate)
        icp.setInputSource(sc)
        icp.setInputTarget(tc)
        icp.begin()
        want_idx, want_d2 = brute(src, tgt, gate)
        for _ in range(2):     # second round: seeded
            idx, d2 = icp.search()
            assert np.array_equal(idx.astype(np.int64), want_idx), (kind, nt, ns, gate, rep)
            assert np.array_equal(d2[want_idx >= 0], want_d2[want_idx >= 0]), (kind, nt, ns, gate, rep)
        icp.end()
        n_checked += ns
    assert n_checked > 0


def test_wide_extent_cloud_takes_the_brick_hash_and_stays_exact(api, rs):
    """Two clusters 400 m apart: the dense table would need ~10^13 cells at this gate, so the
    index falls back to the brick hash by itself (no environment override)."""
    rng = np.random.default_rng(77)
    a = rng.uniform(-0.5, 0.5, (3000, 3))
    b = rng.uniform(-0.5, 0.5, (3000, 3)) + np.array([400.0, -250.0, 100.0])
    tgt = np.concatenate([a, b]).astype(np.float32)
    src = (tgt[rng.permutation(len(tgt))[:2500]] + rng.normal(0, 0.004, (2500, 3))).astype(np.float32)
    gate = 0.02
    icp = api.IterativeClosestPoint()
    icp.params = api.icp_params(max_iterations=2, criteria_mode=1, pipeline_mode=0, max_correspondence_distance=gate)
    icp.setInputSource(rs.PointCloud.from_xyz(src))
    icp.setInputTarget(rs.PointCloud.from_xyz(tgt))
    icp.begin()
    idx, d2 = icp.search()
    assert icp.grid_info().index_kind == 0          # brick hash
    want_idx, want_d2 = brute(src, tgt, gate)
    assert np.array_equal(idx.astype(np.int64), want_idx)
    assert np.array_equal(d2[want_idx >= 0], want_d2[want_idx >= 0])
    icp.end()
    # and a whole alignment in each pipeline on that index
    out = []
    for pipeline in (0, 1, 2):
        icp.params = api.icp_params(max_iterations=4, criteria_mode=1, pipeline_mode=pipeline, max_correspondence_distance=gate)
        icp.align()
        out.append((bytes(icp.result.transform), icp.result.n_correspondences))
    assert out[0] == out[1] == out[2]


def test_tiny_and_identical_clouds(api, rs):
    rng = np.random.default_rng(3)
    pts = rng.uniform(-1, 1, (500, 3)).astype(np.float32)
    # identical clouds: every distance is exactly zero, the transform stays the identity
    icp = api.IterativeClosestPoint()
    icp.params = api.icp_params(max_iterations=3, criteria_mode=1, max_correspondence_distance=0.05)
    icp.setInputSource(rs.PointCloud.from_xyz(pts))
    icp.setInputTarget(rs.PointCloud.from_xyz(pts))
    icp.begin()
    idx, d2 = icp.search()
    assert np.array_equal(idx, np.arange(500)) and not d2.any()
    icp.end()
    icp.align()
    assert np.abs(icp.getFinalTransformation() - np.eye(4)).max() < 1e-6
    # one, two and three points
    for n in (1, 2, 3):
        icp = api.IterativeClosestPoint()
        icp.params = api.icp_params(max_iterations=2, criteria_mode=1, max_correspondence_distance=1.0)
        icp.setInputSource(rs.PointCloud.from_xyz(pts[:n] + 0.001))
        icp.setInputTarget(rs.PointCloud.from_xyz(pts[:n]))
        icp.begin()
        idx, _ = icp.search()
        assert np.array_equal(idx.astype(np.int64), brute(pts[:n] + np.float32(0.001), pts[:n], 1.0)[0])
        icp.end()
        icp.align()
        assert icp.hasConverged() == (n >= 3)        # PCL: fewer than 3 correspondences is a failure


def test_long_grid_many_points_in_one_x_bucket(api, rs):
    """The early exit of an x-sorted walk allows for what the sort order can be off by: one 2^-16 bucket
    of the sort key plus the float rounding of the in-grid position, which grows with the grid (at
    x - x0 = 16..32 m one ulp is 1.9 um).  A 30 m long target with thousands of points whose x differ
    by fractions of an ulp-sized bucket, and queries a hair off them: still the brute-force answer."""
    rng = np.random.default_rng(99)
    n = 6000
    base = np.zeros((n, 3))
    base[:, 0] = 29.5 + rng.integers(0, 40, n) * 1.9e-6          # a few dozen distinct x values, all inside one or two sort buckets
    base[:, 1] = rng.uniform(0, 0.012, n)
    base[:, 2] = rng.uniform(0, 0.012, n)
    anchor = np.array([[0.0, 0.0, 0.0], [30.0, 0.02, 0.02]])     # stretches the grid to 30 m
    tgt = np.concatenate([base, anchor]).astype(np.float32)
    src = (base[rng.permutation(n)[:1500]] + rng.normal(0, 4e-6, (1500, 3)) + np.array([0.0, 1e-4, -1e-4])).astype(np.float32)
    for gate in (0.01, 0.0005):
        icp = api.IterativeClosestPoint()
        icp.params = api.icp_params(max_iterations=2, criteria_mode=1, pipeline_mode=0, max_correspondence_distance=gate)
        icp.setInputSource(rs.PointCloud.from_xyz(src))
        icp.setInputTarget(rs.PointCloud.from_xyz(tgt))
        icp.begin()
        want_idx, want_d2 = brute(src, tgt, gate)
        for _ in range(2):
            idx, d2 = icp.search()
            if not os.environ.get("RSREG_FORCE_HASH") and not os.environ.get("RSREG_DENSE_MAX_CELLS"):
                assert icp.grid_info().index_kind == 1           # the dense table, the index with the x-sorted walks
            assert np.array_equal(idx.astype(np.int64), want_idx)
            assert np.array_equal(d2[want_idx >= 0], want_d2[want_idx >= 0])
        icp.end()
