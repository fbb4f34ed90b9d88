"""Seeded synthetic clouds of D435i-like shape.

An organized depth image rendered through a pinhole of a room scene (back wall, floor, two
side walls, 6 boxes, 3 spheres), with depth noise sigma_z = 0.002 z^2, 1 mm depth
quantisation and ~12 % invalid pixels emitted as the point (0,0,0) — the RealSense
convention the reference's capture path passes through verbatim
(src/capture_opencv.hpp:146-148).  Frame k is a RE-RENDER from pose T_k (true re-sampling,
not the same points moved).  Presets: "parity" (small motion, inside the reference's 1 cm
gate) and "bench" (larger motion for multi-iteration runs).
"""
import math

import numpy as np

from .cloud import POINT_DTYPE, PointCloud

SCENE_SEED = 20240
SIZES = {"50k": (250, 200), "N300": (640, 480), "N1M": (1250, 800)}
PRESETS = {
    # yaw per frame (deg), translation per frame (m)
    "parity": (0.15, (0.0010, -0.0005, 0.0007)),
    "bench": (1.5, (0.012, -0.006, 0.009)),
}


def frame_pose(k, preset="parity"):
    """Camera-to-world pose of frame k (4x4 float64)."""
    yaw_deg, t = PRESETS[preset]
    a = math.radians(yaw_deg * k)
    c, s = math.cos(a), math.sin(a)
    T = np.eye(4)
    T[:3, :3] = [[c, 0, s], [0, 1, 0], [-s, 0, c]]
    T[:3, 3] = [t[0] * k, t[1] * k, t[2] * k]
    return T


def ground_truth(k_src, k_tgt, preset="parity"):
    """Transform taking frame k_src camera coordinates to frame k_tgt's."""
    return np.linalg.inv(frame_pose(k_tgt, preset)) @ frame_pose(k_src, preset)


def _scene(seed=SCENE_SEED):
    rng = np.random.default_rng(seed)
    boxes = []
    for _ in range(6):
        c = np.array([rng.uniform(-1.1, 1.1), rng.uniform(0.2, 0.7), rng.uniform(0.8, 2.0)])
        h = np.array([rng.uniform(0.08, 0.3), rng.uniform(0.1, 0.35), rng.uniform(0.08, 0.3)])
        c[1] = 0.9 - h[1]  # standing on the floor (y is down)
        boxes.append((c - h, c + h))
    spheres = []
    for _ in range(3):
        c = np.array([rng.uniform(-0.9, 0.9), rng.uniform(-0.6, 0.3), rng.uniform(0.6, 1.8)])
        spheres.append((c, rng.uniform(0.08, 0.22)))
    return boxes, spheres


def _raycast(o, d, boxes, spheres):
    """o (3,), d (N,3) -> t (N,) of the first hit along o + t d (all rays hit a wall)."""
    t_best = np.full(d.shape[0], np.inf)

    def plane(axis, value):
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (value - o[axis]) / d[:, axis]
        t[~(t > 1e-6)] = np.inf
        return t

    for axis, value in ((2, 2.2), (1, 0.9), (0, -1.6), (0, 1.6), (1, -1.5)):
        t_best = np.minimum(t_best, plane(axis, value))
    for lo, hi in boxes:
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = (lo - o) / d
            t2 = (hi - o) / d
        tn = np.nanmax(np.minimum(t1, t2), axis=1)
        tf = np.nanmin(np.maximum(t1, t2), axis=1)
        hit = (tn <= tf) & (tn > 1e-6)
        t_best = np.where(hit & (tn < t_best), tn, t_best)
    for c, r in spheres:
        oc = o - c
        a = np.einsum("ij,ij->i", d, d)
        b = 2.0 * (d @ oc)
        cc = oc @ oc - r * r
        disc = b * b - 4 * a * cc
        ok = disc > 0
        sq = np.sqrt(np.where(ok, disc, 0.0))
        t = (-b - sq) / (2 * a)
        hit = ok & (t > 1e-6)
        t_best = np.where(hit & (t < t_best), t, t_best)
    return t_best


def _smooth_noise(rng, h, w, cell):
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.random((gh, gw))
    ys = np.arange(h) / cell
    xs = np.arange(w) / cell
    y0 = ys.astype(int)
    x0 = xs.astype(int)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def render_frame(k=0, size="N300", preset="parity", noise=True, invalid=True, seed=SCENE_SEED):
    """Organized XYZRGB cloud of frame k in ITS OWN camera coordinates."""
    w, h = SIZES[size] if isinstance(size, str) else size
    fx = fy = 385.0 * (w / 640.0)
    cx, cy = w / 2.0, h / 2.0
    boxes, spheres = _scene(seed)
    T = frame_pose(k, preset)
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    dc = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], axis=-1).reshape(-1, 3)
    dw = dc @ T[:3, :3].T
    z = _raycast(T[:3, 3], dw, boxes, spheres)  # d_cam.z == 1 -> t is the camera depth
    hit_w = T[:3, 3] + dw * z[:, None]

    rng = np.random.default_rng(1000 + k)
    zq = z.copy()
    if noise:
        zq = z + rng.standard_normal(z.shape) * (0.002 * z * z)
    zq = np.round(zq * 1000.0) / 1000.0  # Z16 depth units of 1 mm
    bad = np.zeros(z.shape, bool)
    if invalid:
        blob = _smooth_noise(rng, h, w, max(8, w // 40)).reshape(-1)
        bad |= blob > np.quantile(blob, 0.90)
        zi = z.reshape(h, w)
        disc = np.zeros((h, w), bool)
        disc[:, 1:] |= np.abs(zi[:, 1:] - zi[:, :-1]) > 0.08
        disc[:, :-1] |= disc[:, 1:]
        disc[1:, :] |= np.abs(zi[1:, :] - zi[:-1, :]) > 0.08
        disc[:-1, :] |= disc[1:, :]
        bad |= disc.reshape(-1)
    bad |= ~np.isfinite(zq) | (zq <= 0)

    p = (dc * zq[:, None]).astype(np.float32)
    p[bad] = 0.0
    # procedural checker / stripe texture from the world hit position
    chk = (np.floor(hit_w[:, 0] * 8) + np.floor(hit_w[:, 1] * 8) + np.floor(hit_w[:, 2] * 8)).astype(np.int64) & 1
    stripe = (np.floor(hit_w[:, 0] * 25).astype(np.int64) & 3) == 0
    r = np.where(chk == 1, 200, 60) + np.where(stripe, 40, 0)
    g = np.where(chk == 1, 180, 90)
    b = np.where(stripe, 220, 70)
    rgba = (0xFF << 24) | (r.astype(np.uint32) << 16) | (g.astype(np.uint32) << 8) | b.astype(np.uint32)
    rgba = np.where(bad, np.uint32(0xFF000000), rgba).astype(np.uint32)

    pts = np.zeros(w * h, POINT_DTYPE)
    pts["x"], pts["y"], pts["z"] = p[:, 0], p[:, 1], p[:, 2]
    pts["w"] = 1.0
    pts["rgba"] = rgba
    return PointCloud(pts, width=w, height=h, is_dense=False)


def exact_pair(n=4096, seed=7, T=None, spacing=0.12):
    """KAT family: P on a jittered lattice (min spacing 0.6*spacing), Q = T * P exactly,
    ordered the same, no noise.  With a motion below half the spacing and inside the gate
    the nearest neighbour is the identity mapping and one Umeyama step recovers T."""
    rng = np.random.default_rng(seed)
    m = int(math.ceil(n ** (1.0 / 3.0)))
    g = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij"), -1).reshape(-1, 3)[:n]
    P = ((g - (m - 1) / 2.0) * spacing + (rng.random((n, 3)) - 0.5) * 0.4 * spacing)
    P[:, 2] += 1.5
    P = P.astype(np.float32)
    if T is None:
        T = small_transform(0.1, (0.002, -0.001, 0.0015))
    T = np.asarray(T, np.float64)
    Q = (P.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    return PointCloud.from_xyz(P), PointCloud.from_xyz(Q), T


def small_transform(yaw_deg, t):
    a = math.radians(yaw_deg)
    c, s = math.cos(a), math.sin(a)
    T = np.eye(4)
    T[:3, :3] = [[c, 0, s], [0, 1, 0], [-s, 0, c]]
    T[:3, 3] = t
    return T
