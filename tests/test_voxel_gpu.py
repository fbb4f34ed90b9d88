"""GPU: pcl::ApproximateVoxelGrid on the device (csrc/voxel.hip) against the sequential host
filter (csrc/voxel_host.cpp) and the oracle: the same records in the same order, byte for byte --
on the golden fixture, on synthetic frames with the reference's two leaf sizes (1 cm edge
clouds, PCL's default 1 m in IncrementalICP), on shuffled input (the filter is order-dependent)
and on degenerate clouds."""
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


def same(a, b):
    """Same records in the same order (every field bit for bit; the 12 padding bytes of a
    PointXYZRGB are not data)."""
    if len(a) != len(b):
        return False
    return all(np.array_equal(a.points[f].view(np.uint32), b.points[f].view(np.uint32)) for f in ("x", "y", "z", "w", "rgba"))


def both(api, cloud, leaf):
    out = []
    for ctx in (None, api.default_context()):
        f = api.ApproximateVoxelGrid(ctx)
        f.setLeafSize(*leaf)
        f.setInputCloud(cloud)
        out.append(f.filter())
    return out


def test_matches_host_filter_and_golden(api, rs, golden, orc):
    g = golden("approx_voxel")
    for key_in, key_out, leaf in (("in", "leaf_001", 0.01), ("in", "leaf_1", 1.0), ("wide_in", "wide_leaf_01", 0.1)):
        cloud = rs.PointCloud(g[key_in].copy())
        host, gpu = both(api, cloud, (leaf, leaf, leaf))
        exp = g[key_out]
        assert len(gpu) == len(host) == len(exp)
        assert same(gpu, host)
        for f in ("x", "y", "z", "w", "rgba"):
            np.testing.assert_array_equal(gpu.points[f], exp[f])
        o = orc.approx_voxel_grid(g[key_in], (leaf, leaf, leaf))
        for f in ("x", "y", "z", "w", "rgba"):
            np.testing.assert_array_equal(gpu.points[f], o[f])


@pytest.mark.parametrize("size,leaf", [("50k", 0.01), ("N300", 0.01), ("N300", 1.0), ("N300", 0.05)])
def test_matches_host_filter_on_frames(api, rs, size, leaf):
    cloud = rs.synth.render_frame(2, size, "bench")
    host, gpu = both(api, cloud, (leaf, leaf, leaf))
    assert len(host) > 0 and len(gpu) == len(host)
    assert same(gpu, host)
    # order-dependence: a shuffled cloud gives a different stream, and still the same answer on both
    rng = np.random.default_rng(5)
    sh = rs.PointCloud(cloud.points[rng.permutation(len(cloud))].copy())
    host2, gpu2 = both(api, sh, (leaf, leaf, leaf))
    assert same(gpu2, host2)


def test_degenerate_clouds(api, rs):
    leaf = (0.01, 0.02, 0.03)
    empty = rs.PointCloud.from_xyz(np.zeros((0, 3), np.float32))
    host, gpu = both(api, empty, leaf)
    assert len(host) == len(gpu) == 0
    nan = rs.PointCloud.from_xyz(np.full((100, 3), np.nan, np.float32))
    host, gpu = both(api, nan, leaf)
    assert len(host) == len(gpu) == 0
    one = rs.PointCloud.from_xyz(np.array([[0.1, -0.2, 0.3]], np.float32))
    host, gpu = both(api, one, leaf)
    assert len(gpu) == 1 and same(gpu, host)
    # all points in one voxel (one long run), and negative coordinates
    rng = np.random.default_rng(9)
    blob = rs.PointCloud.from_xyz((rng.uniform(-0.004, -0.001, (5000, 3))).astype(np.float32),
                                  rgba=rng.integers(0, 2**32, 5000, dtype=np.uint32))
    host, gpu = both(api, blob, (0.01, 0.01, 0.01))
    assert len(gpu) == 1 and same(gpu, host)
    mixed = rs.PointCloud.from_xyz(rng.uniform(-2, 2, (20000, 3)).astype(np.float32),
                                   rgba=rng.integers(0, 2**32, 20000, dtype=np.uint32))
    mixed.points["x"][::97] = np.inf
    host, gpu = both(api, mixed, (0.25, 0.5, 0.125))
    assert same(gpu, host)


def blob(rs, rng, n, lo, hi, rgba=None):
    xyz = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    if rgba is None:
        rgba = rng.integers(0, 2**32, n, dtype=np.uint32)
    return rs.PointCloud.from_xyz(xyz, rgba=rgba)


@pytest.mark.parametrize("n", [1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 5119, 5120, 5121, 7168, 7169, 11264, 11265, 19456, 19457, 27648, 40000])
def test_huge_run_lengths_around_the_window_edges(api, rs, n):
    """One voxel, one run: k_vox_huge_runs (runs of 4 096 points and more) scans instead of adding -- a prefix of 1 024 points, then
    windows of 1 024, 2 048, 4 096, 8 192, 8 192 ... points; the lengths sit on and beside every edge."""
    rng = np.random.default_rng(n)
    host, gpu = both(api, blob(rs, rng, n, 0.05, 0.95), (1.0, 1.0, 1.0))
    assert len(host) == 1 and same(gpu, host)


def test_huge_runs_whose_sums_leave_the_easy_cases(api, rs):
    """What the scan's checked assumption has to survive: negative coordinates (the sum's sign), colour sums past 2^24 (every odd
    byte is then exactly half way between two floats: round to even, the step depends on the sum's parity), coordinates of
    very different magnitude in one run, and many runs of very different length in one cloud."""
    rng = np.random.default_rng(77)
    cases = [
        (blob(rs, rng, 150_000, -0.95, -0.05), (1.0, 1.0, 1.0)),
        (blob(rs, rng, 200_000, 0.05, 0.95, rgba=np.full(200_000, 0xFFFFFFFF, np.uint32)), (1.0, 1.0, 1.0)),
        (blob(rs, rng, 200_000, 0.05, 0.95, rgba=rng.choice(np.array([0xFFFFFEFD, 0xFF0100FF, 0x00FDFFFE], np.uint32), 200_000)), (1.0, 1.0, 1.0)),
        (blob(rs, rng, 180_000, 1e-3, 900.0), (1000.0, 1000.0, 1000.0)),
    ]
    # magnitudes from 1e-6 to 1e3 in one run (the sum's binade is far above most of its terms)
    wide = blob(rs, rng, 120_000, 0.0, 1.0)
    for f in ("x", "y", "z"):
        wide.points[f] = (10.0 ** rng.uniform(-6, 3, 120_000)).astype(np.float32)
    cases.append((wide, (2000.0, 2000.0, 2000.0)))
    # zeros and negative zeros among the terms
    zeros = blob(rs, rng, 50_000, 0.0, 0.5)
    zeros.points["x"][::3] = 0.0
    zeros.points["y"][::5] = -0.0
    cases.append((zeros, (1.0, 1.0, 1.0)))
    # eight voxels with 47 .. 120 000 points each, interleaved
    parts, sizes = [], [300, 4000, 4096, 9000, 30_000, 60_000, 120_000, 47]
    for k, m in enumerate(sizes):
        xyz = rng.uniform(0.05, 0.95, (m, 3)).astype(np.float32) + np.array([k & 1, (k >> 1) & 1, (k >> 2) & 1], np.float32) * np.float32(-1.0)
        parts.append((xyz, rng.integers(0, 2**32, m, dtype=np.uint32)))
    xyz = np.concatenate([p[0] for p in parts])
    rgba = np.concatenate([p[1] for p in parts])
    order = rng.permutation(len(xyz))
    cases.append((rs.PointCloud.from_xyz(xyz[order], rgba=rgba[order]), (1.0, 1.0, 1.0)))
    for cloud, leaf in cases:
        host, gpu = both(api, cloud, leaf)
        assert len(host) >= 1 and same(gpu, host)
