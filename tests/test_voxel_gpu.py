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
