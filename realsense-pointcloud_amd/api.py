"""Host-side mirror of the PCL class surface the reference's schemes call, over the C ABI.

Method names and argument meaning follow the reference call sites:
``pcl::IterativeClosestPoint`` (src/incremental_icp.hpp:46-63), ``pcl::NormalDistributions
Transform`` (src/ndt_edge_based_registration.hpp:38-43,71-72,83,104), ``pcl::ApproximateVoxel
Grid`` (src/icp_edge_based_registration.hpp:47,59-60) and ``pcl::transformPointCloud``
(src/incremental_icp.hpp:63).  Transforms cross this layer as 4x4 numpy arrays in the usual
row/column maths convention; the column-major packing of the ABI is handled here.
"""
import ctypes as C

import numpy as np

from . import lib as _l
from .cloud import POINT_DTYPE, PointCloud

CONV_STATES = ["NOT_CONVERGED", "ITERATIONS", "TRANSFORM", "ABS_MSE", "REL_MSE", "NO_CORRESPONDENCES",
               "FAILURE_AFTER_MAX_ITERATIONS"]


def _colmajor(T):
    if T is None:
        return None
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).reshape(4, 4).T).copy()


def _rowmajor(buf):
    return np.array(buf, dtype=np.float32).reshape(4, 4).T.copy()


def _records(a):
    """(keepalive, pointer, n, stride) for a PointCloud / structured array / (n,>=3) float32."""
    if isinstance(a, PointCloud):
        a = a.points
    a = np.ascontiguousarray(a)
    if a.dtype.names:
        return a, a.ctypes.data, a.shape[0], a.dtype.itemsize
    if a.dtype != np.float32 or a.ndim != 2 or a.shape[1] < 3:
        raise ValueError("points must be a PointCloud, a structured array or an (n, >=3) float32 array")
    return a, a.ctypes.data, a.shape[0], a.shape[1] * 4


def device_count():
    n = C.c_int(0)
    _l.check(_l.lib().rsreg_device_count(C.byref(n)))
    return n.value


class Context:
    """One (device, stream) execution context; not thread-safe (include/rsreg.h)."""

    def __init__(self, device=0, stream=None, profiling=False):
        h = C.c_void_p()
        _l.check(_l.lib().rsreg_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self.h = h
        self.device = device
        # a context holds ONE ICP target index, ONE ICP source and ONE NDT voxel grid: who uploaded each last
        self.icp_target_owner = self.icp_source_owner = self.ndt_target_owner = None
        if profiling:
            self.set_profiling(True)

    def set_profiling(self, on):
        _l.check(_l.lib().rsreg_ctx_set_profiling(self.h, int(bool(on))), self.h)

    def synchronize(self):
        _l.check(_l.lib().rsreg_ctx_synchronize(self.h), self.h)

    def prepare(self, frame_bytes=0, model_bytes=0, side_streams=False):
        """What a frame loop is about to need -- streams, pinned staging for frames of `frame_bytes`, a device buffer a model
        can grow to `model_bytes` in -- made on a thread of the context while the caller goes on (rsreg_ctx_prepare)."""
        _l.check(_l.lib().rsreg_ctx_prepare(self.h, int(frame_bytes), int(model_bytes), 1 if side_streams else 0), self.h)

    def wait_downloads(self):
        """Every DeviceCloud.download_async of this context has landed in its host array when this returns."""
        _l.check(_l.lib().rsreg_ctx_wait_downloads(self.h), self.h)

    def close(self):
        if getattr(self, "h", None):
            _l.lib().rsreg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- N-GPU
    def comm_init(self, unique_id, rank, nranks):
        buf = (C.c_uint8 * _l.UNIQUE_ID_BYTES).from_buffer_copy(bytes(unique_id))
        _l.check(_l.lib().rsreg_comm_init(self.h, buf, rank, nranks), self.h)

    def allreduce_f64(self, arr):
        arr = np.ascontiguousarray(arr, np.float64)
        _l.check(_l.lib().rsreg_comm_allreduce_f64(self.h, arr.ctypes.data, arr.size), self.h)
        return arr


def comm_unique_id():
    buf = (C.c_uint8 * _l.UNIQUE_ID_BYTES)()
    _l.check(_l.lib().rsreg_comm_unique_id(buf))
    return bytes(buf)


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class DeviceCloud:
    """A cloud resident in HBM (rsreg_cloud): whole records plus width / height / is_dense.  What the
    reference's frame loop hands from step to step (filter -> align -> transformPointCloud -> operator+,
    icp_edge_based_registration.hpp:75-120) without the records leaving the GPU."""

    def __init__(self, cloud=None, ctx=None):
        self.ctx = ctx or default_context()
        h = C.c_void_p()
        _l.check(_l.lib().rsreg_cloud_create(self.ctx.h, C.byref(h)), self.ctx.h)
        self.h = h
        if cloud is not None:
            self.upload(cloud)

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            _l.lib().rsreg_cloud_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, cloud):
        pts = np.ascontiguousarray(cloud.points)
        _l.check(_l.lib().rsreg_cloud_upload(self.h, pts.ctypes.data, len(pts), pts.dtype.itemsize, cloud.width, cloud.height,
                                             int(cloud.is_dense)), self.ctx.h)
        return self

    def upload_async(self, cloud):
        """upload() that returns once the records are staged: the PCIe copy runs beside the main stream's work and
        whatever touches this cloud next waits for it (rsreg_cloud_upload_async)."""
        pts = np.ascontiguousarray(cloud.points)
        _l.check(_l.lib().rsreg_cloud_upload_async(self.h, pts.ctypes.data, len(pts), pts.dtype.itemsize, cloud.width, cloud.height,
                                                   int(cloud.is_dense)), self.ctx.h)
        return self

    def upload_deferred(self, cloud):
        """upload_async() that returns before the records have been read (rsreg_cloud_upload_deferred): a thread of the
        context stages them and queues their copy.  `cloud.points` must not change until a call that reads or rewrites
        this DeviceCloud has returned (the array itself is kept alive here)."""
        pts = np.ascontiguousarray(cloud.points)
        self._deferred_src = pts
        _l.check(_l.lib().rsreg_cloud_upload_deferred(self.h, pts.ctypes.data, len(pts), pts.dtype.itemsize, cloud.width, cloud.height,
                                                      int(cloud.is_dense)), self.ctx.h)
        return self

    def info(self):
        n, s = C.c_size_t(0), C.c_size_t(0)
        w, h, d = C.c_uint32(0), C.c_uint32(0), C.c_int(0)
        _l.check(_l.lib().rsreg_cloud_info(self.h, C.byref(n), C.byref(s), C.byref(w), C.byref(h), C.byref(d)), self.ctx.h)
        return n.value, s.value, w.value, h.value, bool(d.value)

    def __len__(self):
        return self.info()[0]

    @property
    def stamp(self):
        """(id, version): changes whenever the records in HBM are rewritten (upload, filter, transform, concat, align into it)."""
        import ctypes as C
        i, v = C.c_uint64(), C.c_uint64()
        _l.check(_l.lib().rsreg_cloud_version(self.h, C.byref(i), C.byref(v)))
        return (i.value, v.value)

    @property
    def device_ptr(self):
        """Address of the records in HBM (an int; 0 for a cloud without a buffer)."""
        return int(_l.lib().rsreg_cloud_device_ptr(self.h) or 0)

    def download(self):
        n, stride, w, h, dense = self.info()
        assert stride == POINT_DTYPE.itemsize or n == 0
        pts = np.zeros(n, POINT_DTYPE)
        _l.check(_l.lib().rsreg_cloud_download(self.h, pts.ctypes.data, n), self.ctx.h)
        return PointCloud(pts, width=w, height=h, is_dense=dense)

    def download_async(self, out, offset=0):
        """download() that returns at once (rsreg_cloud_download_async): this cloud's records, as they are when the
        context's stream gets here, go to out[offset:offset + len(self)] (a C-contiguous POINT_DTYPE array the caller
        keeps alive and does not touch until ctx.wait_downloads() has returned); the cloud may be rewritten or dropped
        right away.  Returns the number of records on their way."""
        n = self.info()[0]
        if not (isinstance(out, np.ndarray) and out.dtype == POINT_DTYPE and out.flags.c_contiguous and out.ndim == 1):
            raise ValueError("download_async needs a C-contiguous 1-D POINT_DTYPE array")
        if offset < 0 or offset + n > len(out):
            raise ValueError("download_async: %d records do not fit behind offset %d of %d" % (n, offset, len(out)))
        _l.check(_l.lib().rsreg_cloud_download_async(self.h, out.ctypes.data + offset * POINT_DTYPE.itemsize, n), self.ctx.h)
        return n

    def copy(self):
        out = DeviceCloud(ctx=self.ctx)
        _l.check(_l.lib().rsreg_cloud_copy(self.ctx.h, self.h, out.h), self.ctx.h)
        return out

    def __add__(self, other):
        """PointCloud::operator+ in HBM: self's records followed by other's."""
        out = DeviceCloud(ctx=self.ctx)
        _l.check(_l.lib().rsreg_cloud_concat(self.ctx.h, self.h, other.h, out.h), self.ctx.h)
        return out

    def append(self, other):
        """PointCloud::operator+= in HBM (grows in place; the copy is of `other` only once there is room)."""
        _l.check(_l.lib().rsreg_cloud_concat(self.ctx.h, self.h, other.h, self.h), self.ctx.h)
        return self

    def prepend(self, other):
        """`*self = *other + *self` in HBM (icp_edge_based_registration.hpp:119: the new points go FIRST)."""
        _l.check(_l.lib().rsreg_cloud_concat(self.ctx.h, other.h, self.h, self.h), self.ctx.h)
        return self


def icp_params(reference=False, **kw):
    p = _l.IcpParams()
    (_l.lib().rsreg_icp_params_reference if reference else _l.lib().rsreg_icp_params_default)(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def ndt_params(reference=False, **kw):
    p = _l.NdtParams()
    (_l.lib().rsreg_ndt_params_reference if reference else _l.lib().rsreg_ndt_params_default)(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class IterativeClosestPoint:
    """pcl::IterativeClosestPoint<PointXYZRGB, PointXYZRGB> on the MI355X."""

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()
        self.params = icp_params()
        self.reuse_target_index = False   # engine extra, see _sync_inputs; PCL rebuilds its kd-tree at every setInputTarget
        self._src = self._tgt = None
        self._tgt_dirty = True
        self._src_dirty = True
        self._quiet_search = False   # run_sharded_icp: skip the per-iteration D2H of the correspondences
        self.result = None

    # setters the reference calls (incremental_icp.hpp:46-49)
    def setMaximumIterations(self, n):
        self.params.max_iterations = int(n)

    def setMaxCorrespondenceDistance(self, d):
        if d != self.params.max_correspondence_distance:
            self._tgt_dirty = True  # the grid cell size derives from the gate
        self.params.max_correspondence_distance = float(d)

    def setTransformationEpsilon(self, e):
        self.params.transformation_epsilon = float(e)

    def setTransformationRotationEpsilon(self, e):
        self.params.transformation_rotation_epsilon = float(e)

    def setEuclideanFitnessEpsilon(self, e):
        self.params.euclidean_fitness_epsilon = float(e)

    # optional correspondence filters of pcl::IterativeClosestPoint (off by default and in the reference)
    def setUseReciprocalCorrespondences(self, on):
        self.params.use_reciprocal_correspondences = int(bool(on))

    def setTrimmedRejectorOverlapRatio(self, ratio):
        """addCorrespondenceRejector(CorrespondenceRejectorTrimmed with setOverlapRatio(ratio)); <= 0 or >= 1: none."""
        self.params.trim_overlap_ratio = float(ratio)

    # engine knobs (not in PCL)
    def setCriteriaMode(self, mode):
        self.params.criteria_mode = int(mode)

    def setPipelineMode(self, mode):
        self.params.pipeline_mode = int(mode)

    def setInputSource(self, cloud):
        self._src = cloud
        self._src_dirty = True

    def setInputTarget(self, cloud):
        self._tgt = cloud
        self._tgt_dirty = True  # PCL rebuilds the kd-tree whenever the target is set

    # clouds already resident in HBM (e.g. torch tensors): (device pointer, count, stride)
    def setInputSourceDevice(self, ptr, n, stride):
        self._src = ("device", int(ptr), int(n), int(stride))
        self._src_dirty = True

    def setInputTargetDevice(self, ptr, n, stride):
        self._tgt = ("device", int(ptr), int(n), int(stride))
        self._tgt_dirty = True

    def _sync_inputs(self):
        L, h = _l.lib(), self.ctx.h
        if self._tgt is None or self._src is None:
            raise ValueError("setInputSource / setInputTarget not called")
        # the source first, as the reference does (incremental_icp.hpp:57-58): the library loads it on a stream of its own,
        # beside the target's index build
        # a device cloud rewritten in place since it was loaded (filter(x, x), +=, a transform or an alignment into it) is
        # loaded again, as PCL would see the new points through its pointer
        if isinstance(self._src, DeviceCloud) and getattr(self, "_src_stamp", None) != self._src.stamp:
            self._src_dirty = True
        if isinstance(self._tgt, DeviceCloud) and getattr(self, "_tgt_stamp", None) != self._tgt.stamp:
            self._tgt_dirty = True
        if self._src_dirty or self.ctx.icp_source_owner is not self:
            if isinstance(self._src, DeviceCloud):
                _l.check(L.rsreg_icp_set_source_cloud(h, self._src.h), h)
                n = len(self._src)
                self._src_stamp = self._src.stamp
            elif isinstance(self._src, tuple):
                _, p, n, s = self._src
                _l.check(L.rsreg_icp_set_source_device(h, p, n, s, 0), h)
            else:
                keep, p, n, s = _records(self._src)
                dense = int(getattr(self._src, "is_dense", False))
                _l.check(L.rsreg_icp_set_source(h, p, n, s, dense), h)
            self._n_src = n
            self._src_dirty = False
            self.ctx.icp_source_owner = self
        # the index's cell size derives from the gate: a gate that has changed since the build -- through the setter or by assigning
        # `params` -- means a new index (PCL lets the caller change it between two aligns of the same target)
        if getattr(self, "_tgt_gate", None) != self.params.max_correspondence_distance:
            self._tgt_dirty = True
        if self._tgt_dirty or self.ctx.icp_target_owner is not self:
            self._tgt_gate = self.params.max_correspondence_distance
            if isinstance(self._tgt, DeviceCloud):
                # (reuse_target_index: another ICP object of this context has just built the index of this very cloud)
                if not (self.reuse_target_index and L.rsreg_icp_target_is_cloud(h, self._tgt.h, self.params.max_correspondence_distance)):
                    _l.check(L.rsreg_icp_set_target_cloud(h, self._tgt.h, self.params.max_correspondence_distance), h)
                self._tgt_stamp = self._tgt.stamp
            elif isinstance(self._tgt, tuple):
                _, p, n, s = self._tgt
                _l.check(L.rsreg_icp_set_target_device(h, p, n, s, 0, self.params.max_correspondence_distance), h)
            else:
                keep, p, n, s = _records(self._tgt)
                dense = int(getattr(self._tgt, "is_dense", False))
                _l.check(L.rsreg_icp_set_target(h, p, n, s, dense, self.params.max_correspondence_distance), h)
            self._tgt_dirty = False
            self.ctx.icp_target_owner = self

    def align(self, guess=None):
        """icp.align(out[, guess]): returns the aligned cloud (source colours, xyz <- final * xyz)."""
        self._sync_inputs()
        g = _colmajor(guess)
        res = _l.IcpResult()
        if isinstance(self._src, DeviceCloud):   # the aligned cloud stays in HBM too
            out = DeviceCloud(ctx=self.ctx)
            _l.check(_l.lib().rsreg_icp_align_cloud(self.ctx.h, g.ctypes.data if g is not None else None, C.byref(self.params),
                                                    C.byref(res), out.h), self.ctx.h)
            self.result = res
            return out
        if isinstance(self._src, PointCloud) and len(self._src.points):
            # `output = input` is made inside the call (rsreg_icp_align_records), by the host threads that write the aligned positions
            recs = np.ascontiguousarray(self._src.points)
            out = np.empty(len(recs), recs.dtype)
            _l.check(_l.lib().rsreg_icp_align_records(self.ctx.h, g.ctypes.data if g is not None else None, C.byref(self.params),
                                                      C.byref(res), recs.ctypes.data, out.ctypes.data, out.dtype.itemsize), self.ctx.h)
        else:
            out = self._src.points.copy() if isinstance(self._src, PointCloud) else np.zeros(self._n_src, POINT_DTYPE)
            _l.check(_l.lib().rsreg_icp_align(self.ctx.h, g.ctypes.data if g is not None else None, C.byref(self.params),
                                              C.byref(res), out.ctypes.data, out.dtype.itemsize), self.ctx.h)
        self.result = res
        src = self._src if isinstance(self._src, PointCloud) else None
        return PointCloud(out, width=src.width if src else len(out), height=src.height if src else 1,
                          is_dense=src.is_dense if src else False)

    def hasConverged(self):
        return bool(self.result.converged)

    def getFinalTransformation(self):
        return _rowmajor(self.result.transform)

    def getConvergenceState(self):
        return CONV_STATES[self.result.state]

    # ---- step-wise form (parity tests, sharded runs)
    def begin(self, guess=None):
        self._sync_inputs()
        g = _colmajor(guess)
        _l.check(_l.lib().rsreg_icp_begin(self.ctx.h, g.ctypes.data if g is not None else None, C.byref(self.params)),
                 self.ctx.h)

    def search(self, want_output=True):
        if not want_output:
            _l.check(_l.lib().rsreg_icp_search(self.ctx.h, None, None), self.ctx.h)
            return None
        idx = np.empty(self._n_src, np.int32)
        d2 = np.empty(self._n_src, np.float32)
        _l.check(_l.lib().rsreg_icp_search(self.ctx.h, idx.ctypes.data, d2.ctypes.data), self.ctx.h)
        return idx, d2

    def sums(self):
        s = np.zeros(_l.NUM_SUMS, np.float64)
        _l.check(_l.lib().rsreg_icp_sums(self.ctx.h, s.ctypes.data), self.ctx.h)
        return s

    def update(self, sums):
        sums = np.ascontiguousarray(sums, np.float64)
        t = np.zeros(16, np.float32)
        done = C.c_int(0)
        _l.check(_l.lib().rsreg_icp_update(self.ctx.h, sums.ctypes.data, t.ctypes.data, C.byref(done)), self.ctx.h)
        return _rowmajor(t), bool(done.value)

    def end(self, want_aligned=False):
        res = _l.IcpResult()
        out = np.zeros((self._n_src, 4), np.float32) if want_aligned else None
        _l.check(_l.lib().rsreg_icp_end(self.ctx.h, C.byref(res), out.ctypes.data if want_aligned else None, 16),
                 self.ctx.h)
        self.result = res
        return (res, out) if want_aligned else res

    def grid_info(self):
        gi = _l.GridInfo()
        _l.check(_l.lib().rsreg_icp_grid_info(self.ctx.h, C.byref(gi)), self.ctx.h)
        return gi


class NormalDistributionsTransform:
    """pcl::NormalDistributionsTransform<PointXYZRGB, PointXYZRGB> on the MI355X."""

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()
        self.params = ndt_params()
        self._src = self._tgt = None
        self._tgt_dirty = True
        self.result = None
        self._pcl_centroids = False

    def setPclCentroids(self, on):
        """Engine extra: search the voxels by PCL's own centroid arithmetic (a float running sum per voxel in input order,
        rsreg_ndt_set_centroid_mode) instead of the rounded f64 mean."""
        if bool(on) != self._pcl_centroids:
            self._tgt_dirty = True
        self._pcl_centroids = bool(on)

    def centroids(self):
        n = C.c_int32(0)
        _l.check(_l.lib().rsreg_ndt_get_voxels(self.ctx.h, C.byref(n), None, None, 0), self.ctx.h)
        out = np.zeros((n.value, 3), np.float32)
        _l.check(_l.lib().rsreg_ndt_get_centroids(self.ctx.h, out.ctypes.data, n.value), self.ctx.h)
        return out

    def setTransformationEpsilon(self, e):
        self.params.transformation_epsilon = float(e)

    def setStepSize(self, s):
        self.params.step_size = float(s)

    def setResolution(self, r):
        if r != self.params.resolution:
            self._tgt_dirty = True
        self.params.resolution = float(r)

    def setMaximumIterations(self, n):
        self.params.max_iterations = int(n)

    def setInputSource(self, cloud):
        self._src = cloud

    def setInputTarget(self, cloud):
        self._tgt = cloud
        self._tgt_dirty = True

    def _sync_target(self):
        if self._tgt is None or self._src is None:
            raise ValueError("setInputSource / setInputTarget not called")
        if isinstance(self._tgt, DeviceCloud) and getattr(self, "_tgt_stamp", None) != self._tgt.stamp:
            self._tgt_dirty = True   # (rewritten in place since the voxel grid was built)
        if self._tgt_dirty or self.ctx.ndt_target_owner is not self:
            _l.check(_l.lib().rsreg_ndt_set_centroid_mode(self.ctx.h, 1 if self._pcl_centroids else 0), self.ctx.h)
            if isinstance(self._tgt, DeviceCloud):
                _l.check(_l.lib().rsreg_ndt_set_target_cloud(self.ctx.h, self._tgt.h, self.params.resolution), self.ctx.h)
                self._tgt_stamp = self._tgt.stamp
            else:
                keep, p, n, s = _records(self._tgt)
                _l.check(_l.lib().rsreg_ndt_set_target(self.ctx.h, p, n, s, int(getattr(self._tgt, "is_dense", False)),
                                                       self.params.resolution), self.ctx.h)
            self._tgt_dirty = False
            self.ctx.ndt_target_owner = self

    def align(self, guess=None):
        self._sync_target()
        g = _colmajor(guess)
        res = _l.NdtResult()
        if isinstance(self._src, DeviceCloud):
            out = DeviceCloud(ctx=self.ctx)
            _l.check(_l.lib().rsreg_ndt_align_cloud(self.ctx.h, self._src.h, g.ctypes.data if g is not None else None,
                                                    C.byref(self.params), C.byref(res), out.h), self.ctx.h)
            self.result = res
            return out
        keep, p, n, s = _records(self._src)
        out = self._src.points.copy() if isinstance(self._src, PointCloud) else np.zeros(n, POINT_DTYPE)
        _l.check(_l.lib().rsreg_ndt_align(self.ctx.h, p, n, s, int(getattr(self._src, "is_dense", False)),
                                          g.ctypes.data if g is not None else None, C.byref(self.params),
                                          C.byref(res), out.ctypes.data, out.dtype.itemsize), self.ctx.h)
        self.result = res
        src = self._src if isinstance(self._src, PointCloud) else None
        return PointCloud(out, width=src.width if src else n, height=src.height if src else 1,
                          is_dense=src.is_dense if src else False)

    def hasConverged(self):
        return bool(self.result.converged)

    def getFinalTransformation(self):
        return _rowmajor(self.result.transform)

    def getTransformationProbability(self):
        return self.result.trans_probability

    def derivatives(self, pose):
        self._sync_target()
        keep, p, n, s = _records(self._src)
        pose = np.ascontiguousarray(pose, np.float64)
        score = C.c_double(0)
        g = np.zeros(6)
        h = np.zeros((6, 6))
        _l.check(_l.lib().rsreg_ndt_derivatives(self.ctx.h, p, n, s, 0, pose.ctypes.data, C.byref(score),
                                                g.ctypes.data, h.ctypes.data), self.ctx.h)
        return score.value, g, h

    def voxels(self):
        self._sync_target()
        n = C.c_int32(0)
        _l.check(_l.lib().rsreg_ndt_get_voxels(self.ctx.h, C.byref(n), None, None, 0), self.ctx.h)
        m = np.zeros((n.value, 21), np.float64)
        c = np.zeros(n.value, np.int32)
        _l.check(_l.lib().rsreg_ndt_get_voxels(self.ctx.h, C.byref(n), m.ctypes.data, c.ctypes.data, n.value), self.ctx.h)
        return m, c


class ApproximateVoxelGrid:
    """pcl::ApproximateVoxelGrid<PointXYZRGB>.  Without a context: the sequential host filter
    (csrc/voxel_host.cpp); with one: the GPU filter (csrc/voxel.hip), same records in the same order."""

    def __init__(self, ctx=None):
        self.leaf = np.ones(3, np.float32)  # PCL default leaf: 1 m (IncrementalICP never sets it)
        self._in = None
        self.ctx = ctx

    def setLeafSize(self, lx, ly, lz):
        self.leaf = np.array([lx, ly, lz], np.float32)

    def setInputCloud(self, cloud):
        self._in = cloud

    def filter_async(self):
        """filter() of a device cloud queued by a thread of the context on a stream of its own: returns at once; whatever
        takes the result next waits for its size and its records (rsreg_cloud_filter_async).  The input -- which may be
        the not-yet-complete result of extract_edge_features_async -- must stay alive and unchanged until then."""
        out = DeviceCloud(ctx=self._in.ctx)
        _l.check(_l.lib().rsreg_cloud_filter_async(self._in.ctx.h, self._in.h, self.leaf.ctypes.data, out.h), self._in.ctx.h)
        out._filtered_from = self._in   # (keeps the input alive)
        return out

    def filter(self):
        if isinstance(self._in, DeviceCloud):
            out = DeviceCloud(ctx=self._in.ctx)
            _l.check(_l.lib().rsreg_cloud_filter(self._in.ctx.h, self._in.h, self.leaf.ctypes.data, out.h), self._in.ctx.h)
            return out
        pts = np.ascontiguousarray(self._in.points)
        out = np.zeros_like(pts)
        n_out = C.c_size_t(0)
        if self.ctx is not None:
            _l.check(_l.lib().rsreg_approx_voxel_grid_gpu(self.ctx.h, pts.ctypes.data, len(pts), pts.dtype.itemsize,
                                                          self.leaf.ctypes.data, out.ctypes.data, C.byref(n_out)), self.ctx.h)
        else:
            _l.check(_l.lib().rsreg_approx_voxel_grid(pts.ctypes.data, len(pts), pts.dtype.itemsize,
                                                      self.leaf.ctypes.data, out.ctypes.data, C.byref(n_out)))
        out = out[: n_out.value].copy()
        return PointCloud(out, width=len(out), height=1, is_dense=False)


def transformPointCloud(cloud, T, ctx=None):
    """pcl::transformPointCloud(in, out, Matrix4f): returns the transformed copy."""
    if isinstance(cloud, DeviceCloud):
        out = DeviceCloud(ctx=cloud.ctx)
        t = _colmajor(T)
        _l.check(_l.lib().rsreg_cloud_transform(cloud.ctx.h, cloud.h, t.ctypes.data, out.h), cloud.ctx.h)
        return out
    ctx = ctx or default_context()
    pts = np.ascontiguousarray(cloud.points)
    out = np.empty_like(pts)
    t = _colmajor(T)
    _l.check(_l.lib().rsreg_transform_cloud(ctx.h, pts.ctypes.data, out.ctypes.data, len(pts), pts.dtype.itemsize,
                                            int(cloud.is_dense), t.ctypes.data), ctx.h)
    return PointCloud(out, width=cloud.width, height=cloud.height, is_dense=cloud.is_dense)


def extract_edge_features_async(cloud):
    """extract_edge_features of a DeviceCloud queued by a thread of the context on a stream of its own
    (rsreg_cloud_edge_features_async): returns at once; the result is complete when a call that takes it has waited for
    it (every one does).  `cloud` -- which may still be uploading -- must stay as it is until then."""
    out = DeviceCloud(ctx=cloud.ctx)
    _l.check(_l.lib().rsreg_cloud_edge_features_async(cloud.ctx.h, cloud.h, out.h), cloud.ctx.h)
    out._features_of = cloud   # (keeps the input alive)
    return out


def extract_edge_features(cloud, ctx=None, want_indices=False):
    """extract_edge_features(cloud) of the reference (src/edge_extractor.hpp:7-39): the RGB-Canny edge points
    of an ORGANIZED cloud, in index order.  A DeviceCloud in gives a DeviceCloud out."""
    if isinstance(cloud, DeviceCloud):
        out = DeviceCloud(ctx=cloud.ctx)
        _l.check(_l.lib().rsreg_cloud_edge_features(cloud.ctx.h, cloud.h, out.h), cloud.ctx.h)
        return out
    ctx = ctx or default_context()
    pts = np.ascontiguousarray(cloud.points)
    if cloud.width * cloud.height != len(pts):
        raise ValueError("edge extraction needs an organized cloud (width x height points)")
    out = np.zeros_like(pts)
    idx = np.zeros(len(pts), np.int32)
    n_out = C.c_size_t(0)
    _l.check(_l.lib().rsreg_extract_edge_features(ctx.h, pts.ctypes.data, cloud.width, cloud.height, pts.dtype.itemsize,
                                                  out.ctypes.data, idx.ctypes.data, C.byref(n_out)), ctx.h)
    n = n_out.value
    res = PointCloud(out[:n].copy(), width=n, height=1, is_dense=cloud.is_dense)
    return (res, idx[:n].copy()) if want_indices else res


def umeyama_from_sums(sums):
    sums = np.ascontiguousarray(sums, np.float64)
    t = np.zeros(16, np.float32)
    _l.check(_l.lib().rsreg_umeyama_from_sums(sums.ctypes.data, t.ctypes.data))
    return _rowmajor(t)
