"""GPU: the C ABI takes records of 12, 16 or 32 bytes (packed xyz, xyz + one float, PCL's PointXYZRGB; include/rsreg.h: every
entry point has a `stride`).  The Python layer only ever hands 32-byte records over, so this file goes through ctypes: the same
points in all three layouts must give the same correspondences, the same 4x4 and the same transformed coordinates, bit for
bit -- for host pointers, device pointers and cloud handles (12- and 16-byte clouds take the generic record kernels, 32-byte
ones the two-lanes-per-record form of cloud.hip)."""
import ctypes as C

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


def layouts(cloud):
    """the cloud's xyz as 12-, 16- and 32-byte records (non-finite points kept)"""
    p = cloud.points
    xyz = np.stack([p["x"], p["y"], p["z"]], 1).astype(np.float32)
    r12 = np.ascontiguousarray(xyz)
    r16 = np.zeros((len(xyz), 4), np.float32)
    r16[:, :3] = xyz
    r16[:, 3] = 7.0                     # (something other than the 1.0 an aligned cloud gets)
    return {12: r12, 16: r16, 32: np.ascontiguousarray(p)}


def xyz_of(buf, stride):
    return np.frombuffer(buf.tobytes(), np.float32).reshape(len(buf), stride // 4)[:, :3]


@pytest.mark.parametrize("size", ["50k", "N300"])
def test_three_record_layouts_one_result(api, rs, size):
    from rsreg_amd import lib

    L = lib.lib()
    tgt, src = rs.synth.render_frame(0, size, "parity"), rs.synth.render_frame(1, size, "parity")
    src.points["z"][17] = np.nan
    guess = np.ascontiguousarray(rs.synth.small_transform(0.3, (0.002, -0.001, 0.003)).astype(np.float32).T)
    prm = api.icp_params(max_iterations=3, criteria_mode=1, pipeline_mode=2, max_correspondence_distance=0.03)
    lt, ls = layouts(tgt), layouts(src)
    n = len(src.points)
    got = {}
    for stride in (12, 16, 32):
        t, s = lt[stride], ls[stride]
        for how in ("host", "device", "handles"):
            ctx = api.Context(0)
            res = lib.IcpResult()
            out = np.zeros((n, stride // 4), np.float32) if stride != 32 else np.zeros(n, s.dtype)
            if how == "host":
                lib.check(L.rsreg_icp_set_source(ctx.h, s.ctypes.data, n, stride, 0), ctx.h)
                lib.check(L.rsreg_icp_set_target(ctx.h, t.ctypes.data, len(t), stride, 0, 0.03), ctx.h)
                lib.check(L.rsreg_icp_align(ctx.h, guess.ctypes.data, C.byref(prm), C.byref(res), out.ctypes.data, stride), ctx.h)
                moved = xyz_of(out, stride)
            elif how == "device":
                # (records put into HBM through two handles of ANOTHER context; what the calls get is the bare addresses)
                holder = api.Context(0)
                hd = [C.c_void_p(), C.c_void_p()]
                for h, recs in zip(hd, (t, s)):
                    lib.check(L.rsreg_cloud_create(holder.h, C.byref(h)), holder.h)
                    lib.check(L.rsreg_cloud_upload(h, recs.ctypes.data, len(recs), stride, len(recs), 1, 0), holder.h)
                L.rsreg_cloud_device_ptr.restype = C.c_void_p
                pt, ps = L.rsreg_cloud_device_ptr(hd[0]), L.rsreg_cloud_device_ptr(hd[1])
                lib.check(L.rsreg_icp_set_source_device(ctx.h, C.c_void_p(ps), n, stride, 0), ctx.h)
                lib.check(L.rsreg_icp_set_target_device(ctx.h, C.c_void_p(pt), len(t), stride, 0, 0.03), ctx.h)
                lib.check(L.rsreg_icp_align(ctx.h, guess.ctypes.data, C.byref(prm), C.byref(res), out.ctypes.data, stride), ctx.h)
                moved = xyz_of(out, stride)
                for h in hd:
                    L.rsreg_cloud_destroy(h)
            else:
                hs = [C.c_void_p() for _ in range(3)]
                for h in hs:
                    lib.check(L.rsreg_cloud_create(ctx.h, C.byref(h)), ctx.h)
                ht, hsrc, hout = hs
                lib.check(L.rsreg_cloud_upload(ht, t.ctypes.data, len(t), stride, len(t), 1, 0), ctx.h)
                lib.check(L.rsreg_cloud_upload(hsrc, s.ctypes.data, n, stride, n, 1, 0), ctx.h)
                lib.check(L.rsreg_icp_set_source_cloud(ctx.h, hsrc), ctx.h)
                lib.check(L.rsreg_icp_set_target_cloud(ctx.h, ht, 0.03), ctx.h)
                lib.check(L.rsreg_icp_align_cloud(ctx.h, guess.ctypes.data, C.byref(prm), C.byref(res), hout), ctx.h)
                lib.check(L.rsreg_cloud_download(hout, out.ctypes.data, n), ctx.h)
                moved = xyz_of(out, stride)
                # pcl::transformPointCloud on the handle, in place and into another cloud: the same coordinates
                T = np.ascontiguousarray(np.array(res.transform, np.float32))
                h2 = C.c_void_p()
                lib.check(L.rsreg_cloud_create(ctx.h, C.byref(h2)), ctx.h)
                lib.check(L.rsreg_cloud_transform(ctx.h, hsrc, T.ctypes.data, h2), ctx.h)
                lib.check(L.rsreg_cloud_transform(ctx.h, hsrc, T.ctypes.data, hsrc), ctx.h)
                o2, o3 = np.zeros_like(out), np.zeros_like(out)
                lib.check(L.rsreg_cloud_download(h2, o2.ctypes.data, n), ctx.h)
                lib.check(L.rsreg_cloud_download(hsrc, o3.ctypes.data, n), ctx.h)
                assert np.array_equal(xyz_of(o2, stride).view(np.uint32), moved.view(np.uint32))
                assert np.array_equal(xyz_of(o3, stride).view(np.uint32), moved.view(np.uint32))
                if stride == 16:          # the fourth float travels with the record (transform) or is set to 1 (aligned cloud)
                    assert (np.frombuffer(o2.tobytes(), np.float32).reshape(n, 4)[:, 3] == 7.0).all()
                    assert (np.frombuffer(out.tobytes(), np.float32).reshape(n, 4)[:, 3] == 1.0).all()
                for h in hs + [h2]:
                    L.rsreg_cloud_destroy(h)
            got[(stride, how)] = (bytes(bytearray(res.transform)), int(res.n_correspondences), int(res.iterations), moved.view(np.uint32).copy())
    ref = got[(32, "host")]
    assert ref[1] > 0.5 * n and ref[2] == 3
    for key, val in got.items():
        assert val[0] == ref[0] and val[1:3] == ref[1:3], key
        assert np.array_equal(val[3], ref[3]), key
