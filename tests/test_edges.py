"""The edge extractor, src/edge_extractor.hpp:7-39: of everything that function computes, the cloud it
returns holds only the points labelled EDGELABEL_RGB_CANNY -- pcl::Edge::detectEdgeCanny (thresholds
40 / 100) on the gray image of the organized cloud.  CPU: the C oracle against the independent numpy
restatement (tests/golden/edge_canny.npz).  GPU: the HIP extractor against the oracle and the golden
vectors, index for index and record for record, on host clouds and on clouds resident in HBM."""
import numpy as np
import pytest


def test_oracle_matches_numpy_golden(orc, golden):
    g = golden("edge_canny")
    w, h = g["frame_wh"]
    np.testing.assert_array_equal(orc.edge_features(g["frame"], int(w), int(h)), g["frame_edges"])
    w, h = g["synthetic_wh"]
    idx = orc.edge_features(g["synthetic"], int(w), int(h))
    np.testing.assert_array_equal(idx, g["synthetic_edges"])
    # the fading step: weak rows hang on strong ones and are kept; the same weak rows alone are dropped, so is the weak blob
    assert set(idx // 60) == set(range(1, 39)) and not ((idx // 60 < 12) & (idx % 60 < 12)).any()
    assert len(orc.edge_features(g["weak_only"], int(w), int(h))) == 0


def test_oracle_edge_cases(orc, rs):
    flat = np.zeros(12 * 9, rs.POINT_DTYPE)
    flat["rgba"] = 0xFF808080
    assert len(orc.edge_features(flat, 12, 9)) == 0                   # no gradient, no edge
    for w, h in ((1, 1), (2, 2), (3, 1), (1, 5), (2, 7)):             # no interior pixel: nothing survives the suppression
        img = np.zeros(w * h, rs.POINT_DTYPE)
        img["rgba"] = (0xFF000000 | (np.arange(w * h) * 50 % 256)).astype(np.uint32)
        assert len(orc.edge_features(img, w, h)) == 0
    step = np.zeros(3 * 3, rs.POINT_DTYPE)
    step["rgba"] = np.where(np.arange(9) % 3 == 2, 0xFFFFFFFF, 0xFF000000).astype(np.uint32)
    np.testing.assert_array_equal(orc.edge_features(step, 3, 3), [4])  # the one interior pixel sits on the step


@pytest.fixture(scope="module")
def api(rs):
    from rsreg_amd import api as a, lib
    lib.build()
    if a.device_count() < 1:
        pytest.fail("no HIP device: the product has no CPU fallback")
    return a


@pytest.mark.gpu
def test_gpu_matches_golden_and_oracle(api, orc, rs, golden):
    g = golden("edge_canny")
    for key, edges in (("frame", "frame_edges"), ("synthetic", "synthetic_edges"), ("weak_only", "weak_only_edges")):
        w, h = g["frame_wh"] if key == "frame" else g["synthetic_wh"]
        c = rs.PointCloud(g[key].copy(), width=int(w), height=int(h), is_dense=False)
        out, idx = api.extract_edge_features(c, want_indices=True)
        np.testing.assert_array_equal(idx, g[edges])
        assert (out.width, out.height, out.is_dense) == (len(idx), 1, False)
        for f in ("x", "y", "z", "w", "rgba"):
            np.testing.assert_array_equal(out.points[f].view(np.uint32), c.points[f][idx].view(np.uint32))
    for size in ("50k", "N300", "N1M"):                                # BASELINE frame sizes
        for k in (0, 1):
            c = rs.synth.render_frame(k, size, "bench")
            out, idx = api.extract_edge_features(c, want_indices=True)
            np.testing.assert_array_equal(idx, orc.edge_features(c.points, c.width, c.height))
            dev = api.extract_edge_features(api.DeviceCloud(c)).download()   # the same on a cloud resident in HBM
            assert (dev.width, dev.height) == (len(idx), 1)
            for f in ("x", "y", "z", "w", "rgba"):
                np.testing.assert_array_equal(dev.points[f].view(np.uint32), out.points[f].view(np.uint32))
        assert 0.02 * len(c) < len(idx) < 0.4 * len(c)


@pytest.mark.gpu
def test_gpu_edge_cases(api, rs):
    flat = np.zeros(12 * 9, rs.POINT_DTYPE)
    flat["rgba"] = 0xFF808080
    assert len(api.extract_edge_features(rs.PointCloud(flat, width=12, height=9))) == 0
    assert len(api.extract_edge_features(rs.PointCloud(flat[:0], width=0, height=0))) == 0
    for w, h in ((1, 1), (2, 2), (3, 1), (1, 5)):
        img = np.zeros(w * h, rs.POINT_DTYPE)
        img["rgba"] = (0xFF000000 | (np.arange(w * h) * 50 % 256)).astype(np.uint32)
        assert len(api.extract_edge_features(rs.PointCloud(img, width=w, height=h))) == 0
    with pytest.raises(ValueError):
        api.extract_edge_features(rs.PointCloud(flat, width=len(flat), height=2))      # not organized
