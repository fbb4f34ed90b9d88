"""ctypes binding of oracle/_build/liborc.so (TEST INFRASTRUCTURE ONLY, see __init__)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liborc.so")
_lib = None

NUM_SUMS = 17


def build(force=False):
    """Compile the C restatement with gcc (Makefile in this directory)."""
    srcs = [os.path.join(_HERE, f) for f in ("icp_oracle.c", "ndt_oracle.c", "edge_oracle.c", "rsreg_oracle.h", "orc_linalg.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


class IcpParams(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32), ("criteria_mode", C.c_int32), ("accum_mode", C.c_int32),
        ("nn_mode", C.c_int32), ("dedup_target", C.c_int32), ("num_threads", C.c_int32),
        ("max_correspondence_distance", C.c_double), ("transformation_epsilon", C.c_double),
        ("transformation_rotation_epsilon", C.c_double), ("euclidean_fitness_epsilon", C.c_double),
        ("use_reciprocal", C.c_int32), ("reserved2", C.c_int32), ("trim_overlap_ratio", C.c_double),
    ]

    @classmethod
    def default(cls):
        p = cls()
        lib().orc_icp_params_default(C.byref(p))
        return p

    @classmethod
    def reference(cls):
        p = cls()
        lib().orc_icp_params_reference(C.byref(p))
        return p


class IcpResult(C.Structure):
    _fields_ = [
        ("transform", C.c_float * 16), ("converged", C.c_int32), ("state", C.c_int32),
        ("iterations", C.c_int32), ("reserved", C.c_int32), ("n_correspondences", C.c_uint64),
        ("mse", C.c_double), ("sums_last", C.c_double * NUM_SUMS),
        ("sec_build", C.c_double), ("sec_search", C.c_double), ("sec_total", C.c_double),
    ]

    @property
    def T(self):
        return np.array(self.transform, dtype=np.float32).reshape(4, 4).T.copy()


class NdtParams(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32), ("reserved", C.c_int32),
        ("transformation_epsilon", C.c_double), ("step_size", C.c_double),
        ("resolution", C.c_double), ("outlier_ratio", C.c_double),
    ]

    @classmethod
    def default(cls):
        p = cls()
        lib().orc_ndt_params_default(C.byref(p))
        return p

    @classmethod
    def reference(cls):
        p = cls()
        lib().orc_ndt_params_reference(C.byref(p))
        return p


class NdtResult(C.Structure):
    _fields_ = [
        ("transform", C.c_float * 16), ("converged", C.c_int32), ("iterations", C.c_int32),
        ("trans_probability", C.c_double), ("score", C.c_double), ("n_voxels", C.c_int32),
        ("n_derivative_passes", C.c_int32), ("sec_total", C.c_double),
    ]

    @property
    def T(self):
        return np.array(self.transform, dtype=np.float32).reshape(4, 4).T.copy()


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp, sz, i32, dbl = C.c_void_p, C.c_size_t, C.c_int, C.c_double
        L.orc_icp_create.restype = vp
        L.orc_icp_destroy.argtypes = [vp]
        L.orc_icp_set_target.argtypes = [vp, vp, sz, sz, i32, i32, i32]
        L.orc_icp_set_source.argtypes = [vp, vp, sz, sz, i32]
        L.orc_icp_begin.argtypes = [vp, vp, C.POINTER(IcpParams)]
        L.orc_icp_search.argtypes = [vp, vp, vp]
        L.orc_icp_sums.argtypes = [vp, vp]
        L.orc_icp_update.argtypes = [vp, vp, vp, C.POINTER(C.c_int)]
        L.orc_icp_end.argtypes = [vp, C.POINTER(IcpResult), vp, sz]
        L.orc_icp_align.argtypes = [vp, vp, C.POINTER(IcpParams), C.POINTER(IcpResult), vp, sz]
        L.orc_icp_get_current.argtypes = [vp, vp]
        L.orc_umeyama_from_sums.argtypes = [vp, vp]
        L.orc_mat4_mul.argtypes = [vp, vp, vp]
        L.orc_mat4_mul.restype = None
        L.orc_transform_cloud.argtypes = [vp, vp, sz, sz, i32, vp]
        L.orc_approx_voxel_grid.argtypes = [vp, sz, sz, vp, vp, C.POINTER(sz)]
        L.orc_edge_rgb_canny.argtypes = [vp, sz, C.c_uint32, C.c_uint32, C.c_float, C.c_float, vp, C.POINTER(sz)]
        L.orc_ndt_create.restype = vp
        L.orc_ndt_destroy.argtypes = [vp]
        L.orc_ndt_set_centroid_mode.argtypes = [vp, i32]
        L.orc_ndt_set_centroid_mode.restype = None
        L.orc_ndt_set_target.argtypes = [vp, vp, sz, sz, i32, dbl]
        L.orc_ndt_get_voxels.argtypes = [vp, C.POINTER(C.c_int32), vp, vp, C.c_int32]
        L.orc_ndt_get_centroids.argtypes = [vp, vp, C.c_int32]
        L.orc_ndt_derivatives.argtypes = [vp, vp, sz, sz, i32, vp, C.POINTER(NdtParams), C.POINTER(dbl), vp, vp]
        L.orc_ndt_align.argtypes = [vp, vp, sz, sz, i32, vp, C.POINTER(NdtParams), C.POINTER(NdtResult), vp, sz]
        _lib = L
    return _lib


def _pts(a):
    """(pointer, n, stride) of an array whose records start with float x, y, z."""
    a = np.ascontiguousarray(a)
    if a.dtype.names:
        assert a.ndim == 1
        return a, a.ctypes.data, a.shape[0], a.dtype.itemsize
    assert a.dtype == np.float32 and a.ndim == 2 and a.shape[1] >= 3
    return a, a.ctypes.data, a.shape[0], a.shape[1] * 4


def _mat(T):
    """4x4 row-major numpy -> 16 floats column-major."""
    if T is None:
        return None
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).reshape(4, 4).T).copy()


def _unmat(buf):
    return np.asarray(buf, dtype=np.float32).reshape(4, 4).T.copy()


class IcpOracle:
    def __init__(self):
        self._h = lib().orc_icp_create()
        self.ns = 0

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_icp_destroy(self._h)
            self._h = None

    def set_target(self, pts, is_dense=False, dedup=False, num_threads=1):
        a, p, n, s = _pts(pts)
        rc = lib().orc_icp_set_target(self._h, p, n, s, int(is_dense), int(dedup), num_threads)
        assert rc == 0
        self.nt = n

    def set_source(self, pts, is_dense=False):
        a, p, n, s = _pts(pts)
        assert lib().orc_icp_set_source(self._h, p, n, s, int(is_dense)) == 0
        self.ns = n

    def begin(self, guess, params):
        g = _mat(guess)
        rc = lib().orc_icp_begin(self._h, g.ctypes.data if g is not None else None, C.byref(params))
        assert rc == 0, rc

    def search(self):
        idx = np.empty(self.ns, np.int32)
        d2 = np.empty(self.ns, np.float32)
        assert lib().orc_icp_search(self._h, idx.ctypes.data, d2.ctypes.data) == 0
        return idx, d2

    def sums(self):
        s = np.zeros(NUM_SUMS, np.float64)
        assert lib().orc_icp_sums(self._h, s.ctypes.data) == 0
        return s

    def update(self, sums=None):
        t = np.zeros(16, np.float32)
        done = C.c_int(0)
        sp = None
        if sums is not None:
            sums = np.ascontiguousarray(sums, np.float64)
            sp = sums.ctypes.data
        assert lib().orc_icp_update(self._h, sp, t.ctypes.data, C.byref(done)) == 0
        return _unmat(t), bool(done.value)

    def current(self):
        out = np.empty((self.ns, 3), np.float32)
        lib().orc_icp_get_current(self._h, out.ctypes.data)
        return out

    def end(self, want_aligned=False):
        r = IcpResult()
        out = np.zeros((self.ns, 4), np.float32) if want_aligned else None
        assert lib().orc_icp_end(self._h, C.byref(r), out.ctypes.data if want_aligned else None, 16) == 0
        return (r, out) if want_aligned else r

    def align(self, guess, params, want_aligned=False):
        r = IcpResult()
        g = _mat(guess)
        out = np.zeros((self.ns, 4), np.float32) if want_aligned else None
        rc = lib().orc_icp_align(self._h, g.ctypes.data if g is not None else None, C.byref(params),
                                 C.byref(r), out.ctypes.data if want_aligned else None, 16)
        assert rc == 0, rc
        return (r, out) if want_aligned else r


class NdtOracle:
    def __init__(self):
        self._h = lib().orc_ndt_create()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_ndt_destroy(self._h)
            self._h = None

    def set_centroid_mode(self, mode):
        lib().orc_ndt_set_centroid_mode(self._h, int(mode))

    def set_target(self, pts, resolution, is_dense=False):
        a, p, n, s = _pts(pts)
        assert lib().orc_ndt_set_target(self._h, p, n, s, int(is_dense), float(resolution)) == 0

    def voxels(self):
        n = C.c_int32(0)
        lib().orc_ndt_get_voxels(self._h, C.byref(n), None, None, 0)
        m = np.zeros((n.value, 21), np.float64)
        c = np.zeros(n.value, np.int32)
        lib().orc_ndt_get_voxels(self._h, C.byref(n), m.ctypes.data, c.ctypes.data, n.value)
        return m, c

    def centroids(self):
        n = C.c_int32(0)
        lib().orc_ndt_get_voxels(self._h, C.byref(n), None, None, 0)
        out = np.zeros((n.value, 3), np.float32)
        lib().orc_ndt_get_centroids(self._h, out.ctypes.data, n.value)
        return out

    def derivatives(self, src, pose, params):
        a, p, n, s = _pts(src)
        pose = np.ascontiguousarray(pose, np.float64)
        score = C.c_double(0)
        g = np.zeros(6)
        h = np.zeros((6, 6))
        rc = lib().orc_ndt_derivatives(self._h, p, n, s, 0, pose.ctypes.data, C.byref(params),
                                       C.byref(score), g.ctypes.data, h.ctypes.data)
        assert rc == 0
        return score.value, g, h

    def align(self, src, guess, params, want_aligned=False):
        a, p, n, s = _pts(src)
        g = _mat(guess)
        r = NdtResult()
        out = np.zeros((n, 4), np.float32) if want_aligned else None
        rc = lib().orc_ndt_align(self._h, p, n, s, 0, g.ctypes.data if g is not None else None,
                                 C.byref(params), C.byref(r), out.ctypes.data if want_aligned else None, 16)
        assert rc == 0
        return (r, out) if want_aligned else r


def umeyama_from_sums(sums):
    sums = np.ascontiguousarray(sums, np.float64)
    t = np.zeros(16, np.float32)
    rc = lib().orc_umeyama_from_sums(sums.ctypes.data, t.ctypes.data)
    assert rc == 0
    return _unmat(t)


def mat4_mul(a, b):
    a, b = _mat(a), _mat(b)
    c = np.zeros(16, np.float32)
    lib().orc_mat4_mul(a.ctypes.data, b.ctypes.data, c.ctypes.data)
    return _unmat(c)


def transform_cloud(pts, T, is_dense=False):
    a, p, n, s = _pts(pts)
    out = a.copy()
    t = _mat(T)
    assert lib().orc_transform_cloud(p, out.ctypes.data, n, s, int(is_dense), t.ctypes.data) == 0
    return out


def approx_voxel_grid(pts, leaf):
    a, p, n, s = _pts(pts)
    assert s >= 20
    out = np.zeros_like(a)
    leaf = np.ascontiguousarray(leaf, np.float32)
    n_out = C.c_size_t(0)
    assert lib().orc_approx_voxel_grid(p, n, s, leaf.ctypes.data, out.ctypes.data, C.byref(n_out)) == 0
    return out[: n_out.value].copy()


def edge_features(pts, width, height, t_low=40.0, t_high=100.0):
    """Indices of the RGB-Canny edge points of an organized cloud (src/edge_extractor.hpp:7-39)."""
    a, p, n, s = _pts(pts)
    assert n == width * height and s >= 20
    idx = np.zeros(n, np.int32)
    cnt = C.c_size_t(0)
    assert lib().orc_edge_rgb_canny(p, s, width, height, t_low, t_high, idx.ctypes.data, C.byref(cnt)) == 0
    return idx[: cnt.value].copy()
