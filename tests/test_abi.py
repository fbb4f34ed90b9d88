"""CPU: the C-ABI library builds, loads and exports every symbol include/rsreg.h declares;
host-only entry points behave; without a GPU the compute entry points fail loudly."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L(rs):
    from rsreg_amd import lib
    lib.build()
    return lib


def test_header_and_exports_agree(L):
    hdr = open(os.path.join(ROOT, "include", "rsreg.h")).read()
    declared = set(re.findall(r"^(?:int|void|size_t|const char \*|const void \*)\s*(rsreg_[a-z0-9_]+)\s*\(", hdr, re.M))
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    handle = L.lib()
    for name in L.EXPORTS:
        assert getattr(handle, name) is not None


def test_version_and_status_strings(L):
    assert L.lib().rsreg_version() == 4          # 0.4: rsreg_icp_align_records (0.3: rsreg_grid_info lost n_updates, rsreg_icp_result gained ms_allreduce)
    assert L.status_string(0) == "ok"
    assert "device" in L.status_string(-6)


def test_struct_layouts_match_header(L):
    # sizes the C compiler gives the same structs
    assert C.sizeof(L.IcpParams) == 64
    assert C.sizeof(L.NdtParams) == 40
    assert C.sizeof(L.IcpResult) == 64 + 16 + 8 + 8 + 17 * 8 + 4 * 8 + 8 + 8
    assert C.sizeof(L.GridInfo) == 72


def test_reference_presets(rs, L):
    from rsreg_amd import api
    p = api.icp_params(reference=True)
    assert p.use_reciprocal_correspondences == 0 and p.trim_overlap_ratio == 0.0   # the reference attaches no rejector
    # src/incremental_icp.hpp:46-49
    assert (p.max_iterations, p.max_correspondence_distance, p.transformation_epsilon, p.euclidean_fitness_epsilon) == (100, 0.01, 1.0, 1000.0)
    q = api.ndt_params(reference=True)
    # src/ndt_edge_based_registration.hpp:38-43
    assert (q.transformation_epsilon, q.step_size, q.resolution, q.max_iterations) == (0.01, 0.1, 1.0, 50)
    d = api.icp_params()
    assert d.max_iterations == 10 and d.transformation_epsilon == 0.0


def test_umeyama_from_sums_host(rs, L, golden, orc):
    from rsreg_amd import api
    g = golden("crop_parity")
    T = api.umeyama_from_sums(g["ref_it0_sums"])
    np.testing.assert_allclose(T, g["ref_it0_t_inc"], atol=2e-6)
    np.testing.assert_allclose(T, orc.umeyama_from_sums(g["ref_it0_sums"]), atol=1e-7)


def test_approx_voxel_grid_host_matches_golden(rs, L, golden):
    from rsreg_amd import api
    g = golden("approx_voxel")
    for key_in, key_out, leaf in (("in", "leaf_001", 0.01), ("in", "leaf_1", 1.0), ("wide_in", "wide_leaf_01", 0.1)):
        f = api.ApproximateVoxelGrid()
        if leaf != 1.0:
            f.setLeafSize(leaf, leaf, leaf)
        f.setInputCloud(rs.PointCloud(g[key_in].copy()))
        out = f.filter()
        exp = g[key_out]
        assert len(out) == len(exp) and out.height == 1 and not out.is_dense
        for fld in ("x", "y", "z", "w", "rgba"):
            np.testing.assert_array_equal(out.points[fld], exp[fld])


def test_no_gpu_means_loud_failure(rs, L):
    from rsreg_amd import api
    if api.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(L.RsregError) as e:
        api.Context(0)
    assert e.value.status == -6


def test_product_does_not_import_oracle():
    """The product path must never route through oracle/ (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "realsense-pointcloud_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(base, f), errors="replace").read()
                assert "oracle" not in text.lower() or f in ("icp_kernels.hpp", "host_linalg.hpp"), (base, f)
    for f in ("icp_kernels.hpp", "host_linalg.hpp"):
        text = open(os.path.join(pkg, "csrc", f)).read()
        assert "#include \"../../oracle" not in text and "liborc" not in text
