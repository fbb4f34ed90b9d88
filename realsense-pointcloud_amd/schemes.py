"""The reference's registration schemes over the C ABI (Python mirror of include/rsreg/schemes.hpp).

Same surface and observable behaviour as the reference classes: ``IncrementalICP``
(src/incremental_icp.hpp:33-70), ``ICPEdgeBasedRegistration``
(src/icp_edge_based_registration.hpp:10-136), ``NDTEdgeBasedRegistration``
(src/ndt_edge_based_registration.hpp:7-123) and the two-phase driver (src/types.hpp:22-44).
``extract_features`` is the reference's ``extract_edge_features`` (src/edge_extractor.hpp:7-39: the
RGB-Canny edge points of the organized frame) unless a ``feature_fn`` is plugged in.

The numeric building blocks are injectable (``backend``): the default and only backend in
this package is the HIP engine; tests/ plug a CPU checker into the very same scheme logic to
compare results.

This mirror is SYNCHRONOUS: one step after the other, in the reference's order.  The pipelining of the frame loops (frames
uploaded, filtered and their features extracted ahead; the merged cloud streamed to the host) lives in ONE place, the C++
host layer (include/rsreg/schemes.hpp); what it must not change -- the records and the transforms -- is what
tests/test_schemes_gpu.py compares between the two.
"""
import math

import numpy as np

from .cloud import PointCloud


def rot_x(a):
    c, s = math.cos(a), math.sin(a)
    T = np.eye(4, dtype=np.float32)
    T[1, 1], T[1, 2], T[2, 1], T[2, 2] = c, -s, s, c
    return T


def rot_y(a):
    c, s = math.cos(a), math.sin(a)
    T = np.eye(4, dtype=np.float32)
    T[0, 0], T[0, 2], T[2, 0], T[2, 2] = c, s, -s, c
    return T


def rot_z(a):
    c, s = math.cos(a), math.sin(a)
    T = np.eye(4, dtype=np.float32)
    T[0, 0], T[0, 1], T[1, 0], T[1, 1] = c, -s, s, c
    return T


class HipBackend:
    """The product path: everything through librsreg.so."""

    def __init__(self, ctx=None):
        from . import api
        self.api = api
        self.ctx = ctx

    def icp(self):
        icp = self.api.IterativeClosestPoint(self.ctx)
        icp.setMaximumIterations(100)            # incremental_icp.hpp:46-49 and the edge schemes
        icp.setMaxCorrespondenceDistance(0.01)
        icp.setTransformationEpsilon(1)
        icp.setEuclideanFitnessEpsilon(1000)
        return icp

    def ndt(self):
        ndt = self.api.NormalDistributionsTransform(self.ctx)
        ndt.setTransformationEpsilon(0.01)       # ndt_edge_based_registration.hpp:38-43
        ndt.setStepSize(0.1)
        ndt.setResolution(1.0)
        ndt.setMaximumIterations(50)
        return ndt

    def voxel(self, leaf=None):
        f = self.api.ApproximateVoxelGrid(self.ctx or self.api.default_context())   # the GPU filter (same records as the host one)
        if leaf is not None:
            f.setLeafSize(*leaf)
        return f

    def transform(self, cloud, T):
        return self.api.transformPointCloud(cloud, T, self.ctx)

    def edge_features(self, cloud):
        return self.api.extract_edge_features(cloud, self.ctx)

    # what a scheme does with whole clouds between the steps (host clouds here: nothing to move)
    def upload(self, cloud):
        return cloud

    def download(self, cloud):
        return cloud

    def concat(self, a, b):
        return a + b

    def concat_front(self, new, target):
        """`*target = *new + *target` (icp_edge_based_registration.hpp:119): returns the grown target"""
        return new + target


class HipDeviceBackend(HipBackend):
    """The product path with the frame loop resident in HBM: a frame is uploaded once, every step
    (filter, align, transform, concatenate) takes and leaves its clouds on the GPU, and only what the
    caller gets back is downloaded.  Same records, same order as HipBackend."""

    def upload(self, cloud):
        if isinstance(cloud, self.api.DeviceCloud):   # (already there: the two-phase driver uploads a frame once)
            return cloud
        return self.api.DeviceCloud(cloud, self.ctx or self.api.default_context())

    def download(self, cloud):
        return cloud.download()

    def concat_front(self, new, target):
        return target.prepend(new)   # (in place: one copy of `new` in front, the handle keeps its buffer)


def _assign(dst, src):
    """`*dst = *src` for host clouds: the caller's object takes the new contents."""
    dst.points, dst.width, dst.height, dst.is_dense = src.points, src.width, src.height, src.is_dense
    return dst


class RegistrationScheme:
    # True: the reference's progress lines on stdout, text for text (types.hpp:35-41, icp_edge_based_registration.hpp:27-32,
    # 94-96,103-104,110,113,122,127, ndt_edge_based_registration.hpp:24-29,82-84,91-93,98,101,110,114)
    verbose = False

    def __init__(self, backend=None):
        self.backend = backend or HipDeviceBackend()

    def _say(self, text, end="\n"):
        if self.verbose:
            print(text, end=end, flush=True)

    def registration(self, clouds):
        raise NotImplementedError


class TwoPhaseRegistrationScheme(RegistrationScheme):
    feature_fn = None

    def extract_features(self, cloud):
        return self.feature_fn(cloud) if self.feature_fn else self.backend.edge_features(cloud)

    def global_registration(self, pairs):
        raise NotImplementedError

    def registration(self, clouds):
        # a frame goes to the GPU once: its features are extracted there and both stay there for the frame loop
        # (a plugged-in feature_fn is a host function: it gets, and returns, host clouds)
        if self.feature_fn:
            pairs = []
            for f in clouds:
                self._say("[PCL] Extracting features...", end="")
                pairs.append((self.extract_features(f), f))
                self._say("OK")
            self._say("[PCL] Performing global registration...")
            return self.global_registration(pairs)
        for _ in clouds:   # the same lines in the same order; the features themselves are extracted as each frame reaches the GPU
            self._say("[PCL] Extracting features...OK")
        self._say("[PCL] Performing global registration...")
        return self.global_registration(_FramePairs(self, clouds))


class _FramePairs:
    """The (features, frame) pairs of types.hpp:30-43, made when the frame loop gets to them: a frame goes to the GPU once
    (backend.upload) and its features are extracted there."""

    def __init__(self, scheme, clouds):
        self.scheme, self.clouds = scheme, clouds
        self.pairs = {}

    def __len__(self):
        return len(self.clouds)

    def __getitem__(self, k):
        if k < 0:
            k += len(self.clouds)
        if k not in self.pairs:
            f = self.scheme.backend.upload(self.clouds[k])
            self.pairs[k] = (self.scheme.extract_features(f), f)
        return self.pairs[k]

    def __iter__(self):
        return (self[k] for k in range(len(self)))


class IncrementalICP(RegistrationScheme):
    def registration(self, clouds):
        b = self.backend
        voxel = b.voxel()                 # leaf never set -> PCL's 1 m default
        icp = b.icp()
        model = b.upload(clouds[0])       # frame 0 IS the model: it grows (incremental_icp.hpp:40,64)
        self.transforms = []
        self.merged_frames = []           # indices of the frames whose alignment converged (the others are skipped)
        for k in range(1, len(clouds)):
            frame = b.upload(clouds[k])
            voxel.setInputCloud(frame)
            reduced = voxel.filter()
            icp.setInputSource(reduced)
            icp.setInputTarget(model)
            icp.align()
            if not icp.hasConverged():
                continue
            moved = b.transform(frame, icp.getFinalTransformation())
            model = model.append(moved) if hasattr(model, "append") else b.concat(model, moved)
            self.transforms.append(icp.getFinalTransformation())
            self.merged_frames.append(k)
        # the caller's frame 0 has become the merged cloud
        return _assign(clouds[0], b.download(model))


class _EdgeBased(TwoPhaseRegistrationScheme):
    coarse_name = "ICP"
    has_byproducts = False
    # ICPEdgeBasedRegistration writes files while it runs (icp_edge_based_registration.hpp:66-69,126): every frame's edge
    # cloud as <dir>/edge-<k>.pcd (frame 0's already voxel-filtered, the others as extracted) and the grown edge target as
    # <dir>/edge_cloud.pcd, all with savePCDFileBinary.  Opt-in; the directory must exist (the reference's "dataset").
    write_byproducts = False
    byproduct_dir = "dataset"

    def _save(self, name, cloud):
        import os

        from .cloud import save_pcd
        save_pcd(os.path.join(self.byproduct_dir, name), cloud if isinstance(cloud, PointCloud) else self.backend.download(cloud), binary=True)

    def __init__(self, thetas=None, rads=-0.523599, backend=None):
        super().__init__(backend)
        self.thetas = None if thetas is None else [list(map(float, t)) for t in thetas]
        self.use_imu = thetas is not None
        self.rads = float(rads)

    def _coarse(self):
        raise NotImplementedError

    def _imu_guess(self, theta):
        raise NotImplementedError

    def global_registration(self, pairs):
        b = self.backend
        self._say("[PCL] Performing edge-based registration with %s initial rotation guesses..." % ("dynamic" if self.use_imu else "static"))
        by = self.write_byproducts and self.has_byproducts
        if self.use_imu:
            assert len(pairs) == len(self.thetas)
        icp = b.icp()
        if hasattr(icp, "reuse_target_index"):
            icp.reuse_target_index = True    # the coarse ICP of the ICP scheme has just built the index of the same target
        voxel = b.voxel((0.01, 0.01, 0.01))
        coarse = self._coarse()
        merged = b.upload(pairs[0][1])
        voxel.setInputCloud(b.upload(pairs[0][0]))
        target = voxel.filter()                    # frame-0 features: filtered in place, then grown
        if by:
            self._save("edge-0.pcd", target)       # (the reference writes all edge-k.pcd before the loop; the files are the same)
        acc = np.float32(0.0)
        self.frame_transforms = []
        for k in range(1, len(pairs)):
            voxel.setInputCloud(b.upload(pairs[k][0]))
            reduced = voxel.filter()
            if by:
                self._save("edge-%d.pcd" % k, pairs[k][0])
            if self.use_imu:
                t0 = self.thetas[0]
                self.thetas[k] = [np.float32(self.thetas[k][i]) + np.float32(-1.0) * np.float32(t0[i]) for i in range(3)]
                guess = self._imu_guess(self.thetas[k])
            else:
                acc = np.float32(acc + np.float32(self.rads))
                guess = rot_y(float(acc))
            coarse.setInputSource(reduced)
            coarse.setInputTarget(target)
            self._say("[PCL]   Performing %s iteration [%d]..." % (self.coarse_name, k), end="")
            coarse_out = coarse.align(guess)
            self._say("OK")
            t_coarse = coarse.getFinalTransformation()
            icp.setInputSource(coarse_out)
            icp.setInputTarget(target)
            self._say("[PCL]   Performing ICP iteration [%d]..." % k, end="")
            refined = icp.align()
            if not icp.hasConverged():
                self._say("")
                continue
            self._say("OK")
            moved = b.transform(b.upload(pairs[k][1]), t_coarse)
            moved = b.transform(moved, icp.getFinalTransformation())
            target = b.concat_front(refined, target) if hasattr(b, "concat_front") else b.concat(refined, target)     # new points first
            merged = b.concat(merged, moved)
            self.frame_transforms.append((t_coarse, icp.getFinalTransformation()))
        if isinstance(pairs[0][0], PointCloud):
            _assign(pairs[0][0], b.download(target))   # the caller's frame-0 feature cloud has become the grown target
        if by:
            self._save("edge_cloud.pcd", target)
        self._say("[PCL] Done")
        out = b.download(merged)
        return PointCloud(out.points, width=len(out), height=1, is_dense=out.is_dense)


class ICPEdgeBasedRegistration(_EdgeBased):
    has_byproducts = True

    def _coarse(self):
        return self.backend.icp()

    def _imu_guess(self, t):
        return (rot_z(t[0]) @ rot_y(-t[1]) @ rot_x(t[2])).astype(np.float32)


class NDTEdgeBasedRegistration(_EdgeBased):
    coarse_name = "NDT"

    def _coarse(self):
        return self.backend.ndt()

    def _imu_guess(self, t):
        return rot_y(-t[1])
