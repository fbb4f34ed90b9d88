"""The reference's registration schemes over the C ABI (Python mirror of include/rsreg/schemes.hpp).

Same surface and observable behaviour as the reference classes: ``IncrementalICP``
(src/incremental_icp.hpp:33-70), ``ICPEdgeBasedRegistration``
(src/icp_edge_based_registration.hpp:10-136), ``NDTEdgeBasedRegistration``
(src/ndt_edge_based_registration.hpp:7-123) and the two-phase driver (src/types.hpp:22-44).
The RGB-Canny edge extractor is out of scope: ``extract_features`` calls a
user-supplied function, by default the identity.

The numeric building blocks are injectable (``backend``): the default and only backend in
this package is the HIP engine; tests/ plug a CPU checker into the very same scheme logic to
compare results.
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


class RegistrationScheme:
    def __init__(self, backend=None):
        self.backend = backend or HipBackend()

    def registration(self, clouds):
        raise NotImplementedError


class TwoPhaseRegistrationScheme(RegistrationScheme):
    feature_fn = None

    def extract_features(self, cloud):
        return self.feature_fn(cloud) if self.feature_fn else cloud.copy()

    def global_registration(self, pairs):
        raise NotImplementedError

    def registration(self, clouds):
        pairs = [(self.extract_features(c), c) for c in clouds]
        return self.global_registration(pairs)


class IncrementalICP(RegistrationScheme):
    def registration(self, clouds):
        b = self.backend
        voxel = b.voxel()                 # leaf never set -> PCL's 1 m default
        icp = b.icp()
        model = clouds[0]                 # aliases and grows the caller's frame 0
        self.transforms = []
        self.merged_frames = []           # indices of the frames whose alignment converged (the others are skipped)
        for k in range(1, len(clouds)):
            voxel.setInputCloud(clouds[k])
            reduced = voxel.filter()
            icp.setInputSource(reduced)
            icp.setInputTarget(model)
            icp.align()
            if not icp.hasConverged():
                continue
            moved = b.transform(clouds[k], icp.getFinalTransformation())
            merged = model + moved
            model.points, model.width, model.height, model.is_dense = merged.points, merged.width, merged.height, merged.is_dense
            self.transforms.append(icp.getFinalTransformation())
            self.merged_frames.append(k)
        return model


class _EdgeBased(TwoPhaseRegistrationScheme):
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
        if self.use_imu:
            assert len(pairs) == len(self.thetas)
        icp = b.icp()
        voxel = b.voxel((0.01, 0.01, 0.01))
        coarse = self._coarse()
        target = pairs[0][0]                       # frame-0 features: filtered in place, then grown
        merged = PointCloud() + pairs[0][1]
        voxel.setInputCloud(target)
        f0 = voxel.filter()
        target.points, target.width, target.height, target.is_dense = f0.points, f0.width, f0.height, f0.is_dense
        acc = np.float32(0.0)
        self.frame_transforms = []
        for k in range(1, len(pairs)):
            voxel.setInputCloud(pairs[k][0])
            reduced = voxel.filter()
            if self.use_imu:
                t0 = self.thetas[0]
                self.thetas[k] = [np.float32(self.thetas[k][i]) + np.float32(-1.0) * np.float32(t0[i]) for i in range(3)]
                guess = self._imu_guess(self.thetas[k])
            else:
                acc = np.float32(acc + np.float32(self.rads))
                guess = rot_y(float(acc))
            coarse.setInputSource(reduced)
            coarse.setInputTarget(target)
            coarse_out = coarse.align(guess)
            t_coarse = coarse.getFinalTransformation()
            icp.setInputSource(coarse_out)
            icp.setInputTarget(target)
            refined = icp.align()
            if not icp.hasConverged():
                continue
            moved = b.transform(pairs[k][1], t_coarse)
            moved = b.transform(moved, icp.getFinalTransformation())
            grown = refined + target               # new points first
            target.points, target.width, target.height, target.is_dense = grown.points, grown.width, grown.height, grown.is_dense
            merged = merged + moved
            self.frame_transforms.append((t_coarse, icp.getFinalTransformation()))
        return merged


class ICPEdgeBasedRegistration(_EdgeBased):
    def _coarse(self):
        return self.backend.icp()

    def _imu_guess(self, t):
        return (rot_z(t[0]) @ rot_y(-t[1]) @ rot_x(t[2])).astype(np.float32)


class NDTEdgeBasedRegistration(_EdgeBased):
    def _coarse(self):
        return self.backend.ndt()

    def _imu_guess(self, t):
        return rot_y(-t[1])
