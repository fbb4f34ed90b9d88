#!/usr/bin/env python3
"""Generate tests/golden/*.npz with an INDEPENDENT numpy/scipy implementation.

TEST INFRASTRUCTURE ONLY.  The reference path's arithmetic lives in PCL (absent here, the
reference holds no fixtures — SURVEY.md §8c), so these vectors do NOT come from the
reference: they come from a second, independently written implementation of the same
published algorithms (brute-force float32 nearest neighbour + scipy cKDTree sanity check,
numpy.linalg.svd Umeyama, pure-Python ApproximateVoxelGrid, numpy NDT score with finite-
difference gradient/Hessian, numpy/scipy.ndimage Canny for the edge extractor).  The C oracle (oracle/*.c) and the HIP path are both checked
against them.  Parity with PCL itself stays unpinned.

Run:  python oracle/make_golden.py        (writes tests/golden/, a few seconds)
"""
import math
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsreg_amd  # noqa: E402  (input generator only)
from rsreg_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
f32 = np.float32


# ---------------------------------------------------------------- float32 building blocks
def xform_f32(M, xyz):
    """xyz <- M[:3,:3] xyz + M[:3,3] with the op order ((m0 x + m1 y) + m2 z) + m3, all f32."""
    M = np.asarray(M, f32)
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    out = np.empty_like(xyz)
    for r in range(3):
        out[:, r] = ((M[r, 0] * x + M[r, 1] * y) + M[r, 2] * z) + M[r, 3]
    return out


def mat4_mul_f32(A, B):
    A, B = np.asarray(A, f32), np.asarray(B, f32)
    Cm = np.zeros((4, 4), f32)
    for i in range(4):
        for j in range(4):
            s = A[i, 0] * B[0, j]
            s = f32(s + A[i, 1] * B[1, j])
            s = f32(s + A[i, 2] * B[2, j])
            s = f32(s + A[i, 3] * B[3, j])
            Cm[i, j] = s
    return Cm


def brute_nn_f32(q, t, chunk=512):
    """argmin_j ((dx*dx + dy*dy) + dz*dz) in float32, first index on ties."""
    idx = np.empty(len(q), np.int64)
    d2 = np.empty(len(q), f32)
    for a in range(0, len(q), chunk):
        qq = q[a:a + chunk]
        dx = qq[:, None, 0] - t[None, :, 0]
        dy = qq[:, None, 1] - t[None, :, 1]
        dz = qq[:, None, 2] - t[None, :, 2]
        dd = (dx * dx + dy * dy) + dz * dz
        assert dd.dtype == f32
        j = np.argmin(dd, axis=1)
        idx[a:a + chunk] = j
        d2[a:a + chunk] = dd[np.arange(len(qq)), j]
    return idx, d2


def umeyama_np(P, Q):
    """Eigen::umeyama(src=P, dst=Q, with_scaling=false) in float64."""
    mp, mq = P.mean(0), Q.mean(0)
    sigma = (Q - mq).T @ (P - mp) / len(P)
    U, s, Vt = np.linalg.svd(sigma)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = mq - R @ mp
    return T


def sums17(p, q, d2):
    p64, q64 = p.astype(np.float64), q.astype(np.float64)
    s = np.zeros(17)
    s[0] = len(p)
    s[1:4] = p64.sum(0)
    s[4:7] = q64.sum(0)
    s[7:16] = (q64.T @ p64).reshape(-1)
    s[16] = d2.astype(np.float64).sum()
    return s


def icp_np(src, tgt, guess, max_iter, max_dist, trans_eps, rot_eps, fit_eps, fixed):
    """PCL ICP loop, independent of oracle/*.c."""
    src = src.astype(f32)
    tfin = np.isfinite(tgt).all(1)
    tmap = np.nonzero(tfin)[0]
    t = tgt[tfin].astype(f32)
    sval = np.isfinite(src).all(1)
    final = np.asarray(guess, f32).copy()
    cur = src.copy()
    if not np.array_equal(final, np.eye(4, dtype=f32)):
        cur[sval] = xform_f32(final, src[sval])
    gate = float(max_dist) * float(max_dist)
    prev_mse = sys.float_info.max
    rot_thr = rot_eps if rot_eps > 0 else 1.0 - trans_eps
    trace = []
    it, state, conv = 0, 0, False
    while True:
        idx = np.full(len(src), -1, np.int64)
        d2 = np.zeros(len(src), f32)
        j, dd = brute_nn_f32(cur[sval], t)
        idx[sval], d2[sval] = j, dd
        keep = sval & ~(d2.astype(np.float64) > gate)
        n = int(keep.sum())
        rec = {"index": np.where(keep, tmap[np.clip(idx, 0, None)], -1).astype(np.int32),
               "sqr_dist": d2.copy(), "keep": keep.copy()}
        if n < 3:
            state, conv = 5, False
            trace.append(rec)
            break
        p, q = cur[keep], t[idx[keep]]
        s = sums17(p, q, d2[keep])
        Tinc = umeyama_np(p.astype(np.float64), q.astype(np.float64)).astype(f32)
        rec["sums"], rec["t_inc"] = s, Tinc
        trace.append(rec)
        cur[sval] = xform_f32(Tinc, cur[sval])
        final = mat4_mul_f32(Tinc, final)
        it += 1
        mse = s[16] / s[0]
        if it >= max_iter:
            state, conv = 1, True
            break
        if not fixed:
            cos_angle = 0.5 * (float(Tinc[0, 0]) + float(Tinc[1, 1]) + float(Tinc[2, 2]) - 1.0)
            tsq = float(Tinc[0, 3]) ** 2 + float(Tinc[1, 3]) ** 2 + float(Tinc[2, 3]) ** 2
            if cos_angle >= rot_thr and tsq <= trans_eps:
                state, conv = 2, True
                break
            if abs(mse - prev_mse) < 1e-12:
                state, conv = 3, True
                break
            if abs(mse - prev_mse) / prev_mse < fit_eps:
                state, conv = 4, True
                break
            prev_mse = mse
    return {"final": final, "iterations": it, "state": state, "converged": conv, "trace": trace,
            "mse": mse if it else 0.0}


# ---------------------------------------------------------------- ApproximateVoxelGrid (A.5)
def approx_voxel_py(points, leaf):
    inv = [f32(1.0) / f32(l) for l in leaf]
    hist = {}
    out = []

    def flush(h):
        c = [f32(v) / f32(h["count"]) for v in h["c"]]
        rgb = (int(c[4]) << 16) | (int(c[5]) << 8) | int(c[6])
        out.append((c[0], c[1], c[2], rgb))

    rgba = points["rgba"]
    rgbf = rgba.view(f32)
    for i in range(len(points)):
        x, y, z = points["x"][i], points["y"][i], points["z"][i]
        ix = int(math.floor(f32(x * inv[0])))
        iy = int(math.floor(f32(y * inv[1])))
        iz = int(math.floor(f32(z * inv[2])))
        hsh = (ix * 7171 + iy * 3079 + iz * 4231) & 511
        h = hist.get(hsh)
        if h is not None and h["count"] and (h["ix"], h["iy"], h["iz"]) != (ix, iy, iz):
            flush(h)
            h["count"] = 0
            h["c"] = [f32(0)] * 7
        if h is None:
            h = hist[hsh] = {"count": 0, "c": [f32(0)] * 7}
        h["ix"], h["iy"], h["iz"] = ix, iy, iz
        h["count"] += 1
        v = int(rgba[i])
        scratch = [x, y, z, rgbf[i], f32((v >> 16) & 255), f32((v >> 8) & 255), f32(v & 255)]
        with np.errstate(all="ignore"):
            h["c"] = [f32(a + b) for a, b in zip(h["c"], scratch)]
    for hsh in sorted(hist):
        if hist[hsh]["count"]:
            flush(hist[hsh])
    res = np.zeros(len(out), rsreg_amd.POINT_DTYPE)
    for k, (x, y, z, rgb) in enumerate(out):
        res["x"][k], res["y"][k], res["z"][k], res["w"][k], res["rgba"][k] = x, y, z, 1.0, rgb
    return res


# ---------------------------------------------------------------- NDT (A.6 / A.7)
def ndt_voxels_np(tgt, res):
    inv = f32(1.0) / f32(res)
    fin = np.isfinite(tgt).all(1)
    t = tgt[fin]
    mn, mx = t.min(0), t.max(0)
    min_b = np.floor(mn * inv).astype(np.int64)
    max_b = np.floor(mx * inv).astype(np.int64)
    div = max_b - min_b + 1
    ijk = (np.floor(t * inv) - min_b.astype(f32)).astype(np.int64)
    key = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    vox = []
    for k in np.unique(key):
        pts = t[key == k].astype(np.float64)
        n = len(pts)
        if n < 6:
            continue
        mean = pts.mean(0)
        # sum((x-m)(x-m)^T)/n, then PCL's *= (n-1)/n
        cov = (pts - mean).T @ (pts - mean) / n * ((n - 1.0) / n)
        ev, evec = np.linalg.eigh(cov)
        icov = np.zeros((3, 3))
        if not (ev[0] < 0 or ev[1] < 0 or ev[2] <= 0):
            mn_ev = 0.01 * ev[2]
            if ev[0] < mn_ev:
                ev[0] = mn_ev
                if ev[1] < mn_ev:
                    ev[1] = mn_ev
                cov = evec @ np.diag(ev) @ evec.T
            icov = np.linalg.inv(cov)
        cen = (t[key == k].astype(np.float64).sum(0) / n).astype(f32)
        vox.append((n, mean, cov, icov, cen))
    return vox


def ndt_pose_matrix(p):
    cx, sx = math.cos(p[3]), math.sin(p[3])
    cy, sy = math.cos(p[4]), math.sin(p[4])
    cz, sz = math.cos(p[5]), math.sin(p[5])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rx @ Ry @ Rz
    T[:3, 3] = p[:3]
    return T


def ndt_score_np(src, vox, pose, res, outlier=0.55, neigh=None):
    """score(p) = sum over points and neighbour voxels of -d1 exp(-d2/2 x'^T S^-1 x').
    neigh: fixed (point, voxel) incidence so finite differences do not cross the
    discontinuity of the radius test; None -> computed from pose."""
    c1 = 10.0 * (1 - outlier)
    c2 = outlier / res ** 3
    d3 = -math.log(c2)
    d1 = -math.log(c1 + c2) - d3
    d2 = -2 * math.log((-math.log(c1 * math.exp(-0.5) + c2) - d3) / d1)
    T = ndt_pose_matrix(pose)
    x = src.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    if neigh is None:
        xf = xform_f32(T.astype(f32), src.astype(f32))
        neigh = []
        for (n, mean, cov, icov, cen) in vox:
            d = xf - cen
            dd = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            neigh.append(dd < f32(res * res))
    score = 0.0
    for m, (n, mean, cov, icov, cen) in zip(neigh, vox):
        xm = x[m] - mean
        q = np.einsum("ij,jk,ik->i", xm, icov, xm)
        e = np.exp(-d2 * q / 2)
        ok = ~((d2 * e > 1) | (d2 * e < 0) | np.isnan(e))
        score += float((-d1 * e[ok]).sum())
    return score, neigh


def fd_grad_hess(src, vox, pose, res):
    pose = np.asarray(pose, np.float64)
    s0, neigh = ndt_score_np(src, vox, pose, res)
    f = lambda p: ndt_score_np(src, vox, p, res, neigh=neigh)[0]  # noqa: E731
    h = 1e-5
    g = np.zeros(6)
    H = np.zeros((6, 6))
    for i in range(6):
        e = np.zeros(6)
        e[i] = h
        g[i] = (f(pose + e) - f(pose - e)) / (2 * h)
    hh = 1e-4
    for i in range(6):
        for j in range(i, 6):
            ei = np.zeros(6); ei[i] = hh
            ej = np.zeros(6); ej[j] = hh
            H[i, j] = H[j, i] = (f(pose + ei + ej) - f(pose + ei - ej) - f(pose - ei + ej) + f(pose - ei - ej)) / (4 * hh * hh)
    return s0, g, H


# ---------------------------------------------------------------- cases
def rgb_canny_np(rgba, w, h, t_low=40.0, t_high=100.0):
    """pcl::Edge::detectEdgeCanny on gray = (r + g + b) // 3 (what src/edge_extractor.hpp:7-39 returns the
    points of): float32 shifted-add convolutions with clamped borders, numpy arctan2, non-maximum
    suppression by boolean masks, hysteresis by scipy.ndimage.label -- written independently of
    oracle/edge_oracle.c.  Returns the ascending indices of the edge pixels."""
    from scipy import ndimage
    r, g, b = (rgba >> 16) & 255, (rgba >> 8) & 255, rgba & 255
    gray = ((r.astype(np.int64) + g + b) // 3).astype(f32).reshape(h, w)

    def conv(img, k):
        pad = np.pad(img, 1, mode="edge")
        out = np.zeros_like(img)
        for kr in range(3):
            for kc in range(3):
                out = (out + f32(k[kr][kc]) * pad[kr:kr + h, kc:kc + w]).astype(f32)
        return out

    kg = np.array([[np.exp(-(i * i + j * j) / 2.0) for j in (-1, 0, 1)] for i in (-1, 0, 1)]).astype(f32)   # exp in double, rounded once
    tot = f32(0)
    for v in kg.reshape(-1):
        tot = f32(tot + v)
    sm = conv(gray, (kg / tot).astype(f32))
    gx = conv(sm, [[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]])
    gy = conv(sm, [[-1, -2, -1], [0, 0, 0], [1, 2, 1]])
    mag = np.sqrt(gx * gx + gy * gy).astype(f32)
    # (the angle as the correctly rounded float of the true angle: see oracle/edge_oracle.c on atan2f)
    ang = (np.arctan2(gy.astype(np.float64), gx.astype(np.float64)).astype(f32) * f32(57.29578)).astype(f32)
    d = np.full((h, w), -1)
    d[((ang <= 22.5) & (ang >= -22.5)) | (ang >= 157.5) | (ang <= -157.5)] = 0
    d[(d < 0) & (((ang > 22.5) & (ang < 67.5)) | ((ang < -112.5) & (ang > -157.5)))] = 45
    d[(d < 0) & (((ang >= 67.5) & (ang <= 112.5)) | ((ang <= -67.5) & (ang >= -112.5)))] = 90
    d[(d < 0) & (((ang > 112.5) & (ang < 157.5)) | ((ang < -22.5) & (ang > -67.5)))] = 135
    mx = np.zeros((h, w), f32)
    if h > 2 and w > 2:
        nb = {0: ((0, -1), (0, 1)), 45: ((-1, -1), (1, 1)), 90: ((-1, 0), (1, 0)), 135: ((-1, 1), (1, -1))}
        m = mag[1:h - 1, 1:w - 1]
        for k, ((a0, a1), (b0, b1)) in nb.items():
            A = mag[1 + a0:h - 1 + a0, 1 + a1:w - 1 + a1]
            B = mag[1 + b0:h - 1 + b0, 1 + b1:w - 1 + b1]
            sel = (d[1:h - 1, 1:w - 1] == k) & (m >= t_low) & (m >= A) & (m >= B)
            mx[1:h - 1, 1:w - 1][sel] = m[sel]
    lab, _ = ndimage.label(mx > 0, structure=np.ones((3, 3)))
    strong = np.unique(lab[mx >= t_high])
    keep = np.isin(lab, strong[strong > 0])
    return np.nonzero(keep.reshape(-1))[0].astype(np.int32)


def pack_trace(res, prefix, out, max_keep=3):
    for k, rec in enumerate(res["trace"][:max_keep]):
        out["%s_it%d_index" % (prefix, k)] = rec["index"]
        out["%s_it%d_sqr_dist" % (prefix, k)] = rec["sqr_dist"]
        if "sums" in rec:
            out["%s_it%d_sums" % (prefix, k)] = rec["sums"]
            out["%s_it%d_t_inc" % (prefix, k)] = rec["t_inc"]
    out[prefix + "_final"] = res["final"]
    out[prefix + "_meta"] = np.array([res["iterations"], res["state"], int(res["converged"])], np.int64)
    out[prefix + "_mse"] = np.array([res["mse"]])


def kdtree_sanity(q, t, idx, d2):
    """scipy cKDTree (float64) must agree with the float32 brute force except on near-ties."""
    dd, j = cKDTree(t.astype(np.float64)).query(q.astype(np.float64))
    diff = j != idx
    if diff.any():
        assert np.allclose(dd[diff] ** 2, d2[diff].astype(np.float64), rtol=1e-4, atol=1e-12)
    return int(diff.sum())


def main():
    os.makedirs(OUT, exist_ok=True)
    I4 = np.eye(4, dtype=f32)

    # 1. analytic KAT: Q = T P exactly, identity correspondence, one Umeyama step recovers T
    P, Q, T = synth.exact_pair(n=2048, seed=7, T=synth.small_transform(0.1, (0.002, -0.001, 0.0015)))
    r = icp_np(P.xyz, Q.xyz, I4, 100, 0.01, 1.0, 0.0, 1000.0, False)
    out = {"src": P.points, "tgt": Q.points, "T_true": T, "guess": I4}
    pack_trace(r, "ref", out)
    assert (r["trace"][0]["index"] == np.arange(2048)).all()
    assert np.abs(r["final"] - T).max() < 2e-6
    np.savez_compressed(os.path.join(OUT, "kat_exact.npz"), **out)

    # 2. D435i-like crops, reference parameters (one iteration) and a fixed-iteration run
    f0 = synth.render_frame(0, "N300", "parity").crop(300, 200, 64, 64)
    f1 = synth.render_frame(1, "N300", "parity").crop(300, 200, 64, 64)
    out = {"src": f1.points, "tgt": f0.points, "guess": I4}
    r = icp_np(f1.xyz, f0.xyz, I4, 100, 0.01, 1.0, 0.0, 1000.0, False)
    pack_trace(r, "ref", out)
    tfin = f0.xyz[np.isfinite(f0.xyz).all(1)]
    nd = kdtree_sanity(f1.xyz, tfin, brute_nn_f32(f1.xyz, tfin)[0], brute_nn_f32(f1.xyz, tfin)[1])
    r2 = icp_np(f1.xyz, f0.xyz, I4, 8, 0.02, 1e-12, 0.0, 1e-12, True)
    pack_trace(r2, "fixed8", out)
    out["n_zero_src"] = np.array([(f1.points["z"] == 0).sum()])
    np.savez_compressed(os.path.join(OUT, "crop_parity.npz"), **out)
    print("crop_parity: corr", int(r["trace"][0]["keep"].sum()), "kdtree near-tie diffs", nd,
          "zero pts", int(out["n_zero_src"][0]))

    # 3. bench-preset crops with a non-identity guess, PCL criteria with real epsilons
    g0 = synth.render_frame(0, "N300", "bench").crop(200, 150, 80, 60)
    g2 = synth.render_frame(1, "N300", "bench").crop(200, 150, 80, 60)
    guess = synth.small_transform(1.2, (0.01, -0.004, 0.006)).astype(f32)
    out = {"src": g2.points, "tgt": g0.points, "guess": guess}
    r = icp_np(g2.xyz, g0.xyz, guess, 30, 0.05, 1e-9, 0.0, 1e-7, False)
    pack_trace(r, "pcl", out)
    np.savez_compressed(os.path.join(OUT, "crop_bench.npz"), **out)
    print("crop_bench: iterations", r["iterations"], "state", r["state"])

    # 4. ApproximateVoxelGrid, leaf 1 cm (edge schemes) and the default 1 m (IncrementalICP)
    out = {"in": f0.points}
    out["leaf_001"] = approx_voxel_py(f0.points, (0.01, 0.01, 0.01))
    out["leaf_1"] = approx_voxel_py(f0.points, (1.0, 1.0, 1.0))
    wide = synth.render_frame(0, "N300", "parity").crop(0, 0, 640, 480, step=8)
    out["wide_in"] = wide.points
    out["wide_leaf_01"] = approx_voxel_py(wide.points, (0.1, 0.1, 0.1))
    np.savez_compressed(os.path.join(OUT, "approx_voxel.npz"), **out)
    print("voxel: 1cm", len(out["leaf_001"]), "1m", len(out["leaf_1"]), "wide 10cm", len(out["wide_leaf_01"]))

    # 5. NDT: voxel statistics + score / finite-difference gradient and Hessian at a pose
    tgt = synth.render_frame(0, "N300", "bench").crop(0, 0, 640, 480, step=8)
    src = synth.render_frame(1, "N300", "bench").crop(0, 0, 640, 480, step=8)
    keep_t = tgt.points["z"] != 0
    keep_s = src.points["z"] != 0
    tx, sx = tgt.xyz[keep_t], src.xyz[keep_s]
    vox = ndt_voxels_np(tx, 1.0)
    pose = np.array([0.01, -0.005, 0.008, 0.002, 0.03, -0.001])
    s0, g, H = fd_grad_hess(sx, vox, pose, 1.0)
    out = {"tgt": tx, "src": sx, "pose": pose, "score": np.array([s0]), "grad_fd": g, "hess_fd": H,
           "vox_n": np.array([v[0] for v in vox]), "vox_mean": np.array([v[1] for v in vox]),
           "vox_cov": np.array([v[2] for v in vox]), "vox_icov": np.array([v[3] for v in vox])}
    np.savez_compressed(os.path.join(OUT, "ndt_small.npz"), **out)
    print("ndt: voxels", len(vox), "score", s0)

    # 6. edge extractor (RGB Canny of an organized cloud): a 160 x 120 frame, and two hand-made images: a step
    # whose contrast fades along the edge from strong to weak (the hysteresis keeps the weak rows, they hang
    # on the strong ones) beside a weak isolated blob (dropped); and the weak rows of that step alone (dropped)
    c = synth.render_frame(1, (160, 120), "bench")
    idx = rgb_canny_np(c.points["rgba"], 160, 120)

    def fading_step(from_row):
        img = np.full((40, 60), 90, np.int64)
        for r in range(from_row, 40):
            img[r, 30:] = 200 - (5 * r) // 2
        img[5:9, 5:9] = 104
        rec = np.zeros(img.size, rsreg_amd.POINT_DTYPE)
        rec["rgba"] = (0xFF000000 | (img.reshape(-1) << 16) | (img.reshape(-1) << 8) | img.reshape(-1)).astype(np.uint32)
        rec["w"] = 1.0
        return rec

    syn, weak = fading_step(0), fading_step(31)
    idx2, idx3 = rgb_canny_np(syn["rgba"], 60, 40), rgb_canny_np(weak["rgba"], 60, 40)
    assert len(idx3) == 0 and set(idx2 // 60) == set(range(1, 39))
    out = {"frame": c.points, "frame_wh": np.array([160, 120]), "frame_edges": idx, "synthetic": syn, "synthetic_wh": np.array([60, 40]),
           "synthetic_edges": idx2, "weak_only": weak, "weak_only_edges": idx3}
    np.savez_compressed(os.path.join(OUT, "edge_canny.npz"), **out)
    print("edges: frame", len(idx), "of", len(c), "; fading step", len(idx2), "; its weak rows alone", len(idx3))


if __name__ == "__main__":
    main()
