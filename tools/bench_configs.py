#!/usr/bin/env python3
"""Secondary measurements on ONE MI355X for the BASELINE.json configs that are not bench.py's
headline line (those are parity-test cases first; this prints their timings as JSON lines):

  configs[1]  single ICP pair, 2 x 300k synthetic clouds, 30 fixed iterations
  configs[2]  NDT-then-ICP pair: NDT on a ~30k-point subset gives the guess, full-cloud ICP refines
  configs[4]  chain of 16 synthetic 300k frames as independent consecutive pairs (one GPU here)
  reference-parity mode: 1M pair with the reference's parameters (1 iteration, 1 cm gate)

Each line also carries the CPU oracle's time for the same work (single thread) and the
Frobenius distance between the two transforms.  Run: python tools/bench_configs.py
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (CPU side-by-side only)
import rsreg_amd  # noqa: E402
from rsreg_amd import api, synth  # noqa: E402


def timed(fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    t = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return (time.perf_counter() - t) / reps, r


def subset(cloud, step):
    c = cloud.crop(0, 0, cloud.width, cloud.height, step=step)
    pts = np.ascontiguousarray(c.points[c.points["z"] != 0])
    return rsreg_amd.PointCloud(pts, width=len(pts), height=1, is_dense=False)


def main():
    ctx = api.Context(0, profiling=True)
    guess = synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)

    # ---- configs[1]
    tgt, src = synth.render_frame(0, "N300", "bench"), synth.render_frame(1, "N300", "bench")
    icp = api.IterativeClosestPoint(ctx)
    icp.params = api.icp_params(max_iterations=30, criteria_mode=1, max_correspondence_distance=0.05)

    def pair():
        icp.setInputSource(src)
        icp.setInputTarget(tgt)
        icp.align(guess)
        return icp.getFinalTransformation()

    s, T = timed(pair)
    o = oracle.IcpOracle()
    p = oracle.IcpParams.default()
    p.max_iterations, p.criteria_mode, p.max_correspondence_distance = 30, 1, 0.05
    t0 = time.perf_counter()
    o.set_target(tgt.points, dedup=True)
    o.set_source(src.points)
    r = o.align(guess, p)
    cpu = time.perf_counter() - t0
    print(json.dumps({"config": "configs[1] ICP pair 2x300k, 30 iterations, host clouds in", "gpu_ms": s * 1e3,
                      "point_pairs_per_s": len(src) * 30 / s, "nn_kernel_ms_avg": icp.result.ms_nn / icp.result.n_nn_launches,
                      "cpu_port_1thread_ms": cpu * 1e3, "speedup": cpu / s, "T_frobenius_vs_cpu": float(np.linalg.norm(T - r.T))}))

    # ---- configs[2]
    e_t, e_s = subset(tgt, 3), subset(src, 3)
    ndt = api.NormalDistributionsTransform(ctx)
    ndt.params = api.ndt_params(reference=True)
    icp2 = api.IterativeClosestPoint(ctx)
    icp2.params = api.icp_params(max_iterations=30, criteria_mode=1, max_correspondence_distance=0.05)

    def ndt_then_icp():
        ndt.setInputSource(e_s)
        ndt.setInputTarget(e_t)
        ndt.align(synth.small_transform(1.0, (0, 0, 0)).astype(np.float32))
        g = ndt.getFinalTransformation()
        icp2.setInputSource(src)
        icp2.setInputTarget(tgt)
        icp2.align(g)
        return g, icp2.getFinalTransformation()

    s, (Tn, Ti) = timed(ndt_then_icp, reps=3)
    on = oracle.NdtOracle()
    on.set_centroid_mode(1)
    t0 = time.perf_counter()
    on.set_target(e_t.points, 1.0)
    rn = on.align(e_s.points, synth.small_transform(1.0, (0, 0, 0)).astype(np.float32), oracle.NdtParams.reference())
    o.set_target(tgt.points, dedup=True)
    o.set_source(src.points)
    ri = o.align(rn.T, p)
    cpu = time.perf_counter() - t0
    print(json.dumps({"config": "configs[2] NDT (%d-pt subset) then full-cloud ICP (300k, 30 it)" % len(e_s), "gpu_ms": s * 1e3,
                      "ndt_iterations": ndt.result.iterations, "ndt_passes": ndt.result.n_derivative_passes,
                      "ndt_pass_ms_avg": ndt.result.ms_derivatives / max(ndt.result.n_derivative_passes, 1),
                      "cpu_port_1thread_ms": cpu * 1e3, "speedup": cpu / s,
                      "T_ndt_frobenius_vs_cpu": float(np.linalg.norm(Tn - rn.T)), "T_icp_frobenius_vs_cpu": float(np.linalg.norm(Ti - ri.T))}))

    # ---- configs[4]: independent consecutive pairs (k, k-1); the guess is the true frame-to-frame
    # motion perturbed by 0.5 deg / ~8 mm, as a pose prior would be; errors are per pair
    frames = [synth.render_frame(k, "N300", "bench") for k in range(16)]
    chain = api.IterativeClosestPoint(ctx)
    chain.params = api.icp_params(max_iterations=30, criteria_mode=1, max_correspondence_distance=0.05)
    perturb = synth.small_transform(0.5, (0.005, -0.004, 0.005))
    guesses = [(perturb @ synth.ground_truth(k, k - 1, "bench")).astype(np.float32) for k in range(1, 16)]
    err_guess, err_icp = [], []
    t0 = time.perf_counter()
    for k in range(1, 16):
        chain.setInputSource(frames[k])
        chain.setInputTarget(frames[k - 1])
        chain.align(guesses[k - 1])
        Tk = chain.getFinalTransformation().astype(np.float64)
        gt = synth.ground_truth(k, k - 1, "bench")
        err_icp.append(float(np.linalg.norm(Tk - gt)))
        err_guess.append(float(np.linalg.norm(guesses[k - 1] - gt)))
    s = time.perf_counter() - t0
    print(json.dumps({"config": "configs[4] chain of 16 x 300k frames as 15 consecutive pairs, 30 it each, 1 GPU", "gpu_ms": s * 1e3,
                      "point_pairs_per_s": 15 * len(frames[0]) * 30 / s,
                      "pair_pose_error_frobenius_mean": float(np.mean(err_icp)), "pair_pose_error_frobenius_max": float(np.max(err_icp)),
                      "pair_guess_error_frobenius_mean": float(np.mean(err_guess))}))

    # ---- reference-parity mode at 1M
    tgt, src = synth.render_frame(0, "N1M", "parity"), synth.render_frame(1, "N1M", "parity")
    ref = api.IterativeClosestPoint(ctx)
    ref.params = api.icp_params(reference=True)

    def ref_pair():
        ref.setInputSource(src)
        ref.setInputTarget(tgt)
        ref.align()
        return ref.getFinalTransformation()

    s, T = timed(ref_pair)
    t0 = time.perf_counter()
    o.set_target(tgt.points, dedup=True)
    o.set_source(src.points)
    r = o.align(None, oracle.IcpParams.reference())
    cpu = time.perf_counter() - t0
    print(json.dumps({"config": "reference parameters (1 iteration, 1 cm gate), 1M pair, host clouds in", "gpu_ms": s * 1e3,
                      "iterations": ref.result.iterations, "nn_kernel_ms": ref.result.ms_nn, "cpu_port_1thread_ms": cpu * 1e3,
                      "speedup": cpu / s, "T_frobenius_vs_cpu": float(np.linalg.norm(T - r.T))}))


if __name__ == "__main__":
    main()
