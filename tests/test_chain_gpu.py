"""GPU: a chain of frames as consecutive pairs with several pairs IN FLIGHT on one GPU (BASELINE configs[4];
rsreg::ChainRegistrar in include/rsreg/schemes.hpp, its Python mirror in rsreg_amd/chain.py, `bench.py --workload chain
--in-flight K`).  The bar: every pair's 4x4 is BIT-IDENTICAL to the one the pair gets alone on one context -- nothing of an
alignment may depend on what else the GPU is doing -- and equal to the CPU oracle's within the parity tolerance."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_FRAMES = 6


@pytest.fixture(scope="module")
def env(rs):
    from rsreg_amd import api, chain, lib
    lib.build()
    if api.device_count() < 1:
        pytest.fail("no HIP device: the product has no CPU fallback")
    return api, chain, lib


@pytest.fixture(scope="module")
def frames(rs):
    return [rs.synth.render_frame(k, "50k", "bench") for k in range(N_FRAMES)]


def _sequential(api, frames, params, guesses):
    """every pair alone, one after the other, on one context through the cloud handles (the path of rounds 2-5)"""
    ctx = api.Context(0)
    dev = [api.DeviceCloud(f, ctx) for f in frames]
    out = {}
    for k in range(1, len(frames)):
        icp = api.IterativeClosestPoint(ctx)
        icp.params = params
        icp.setInputSource(dev[k])
        icp.setInputTarget(dev[k - 1])
        icp.align(guesses[k] if guesses else None)
        out[k] = (icp.getFinalTransformation().copy(), icp.result.iterations, bool(icp.hasConverged()), int(icp.result.n_correspondences))
    return out


@pytest.mark.parametrize("mode", ["reference", "bench"])
def test_pairs_in_flight_are_bit_identical_to_pairs_alone(env, frames, rs, mode):
    api, chain, lib = env
    if mode == "reference":
        params, guesses = api.icp_params(reference=True), None          # incremental_icp.hpp:46-49, identity guess (:59)
    else:
        params = api.icp_params(max_iterations=12, criteria_mode=1, pipeline_mode=2, max_correspondence_distance=0.05)
        guesses = {k: rs.synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32) for k in range(1, N_FRAMES)}
    alone = _sequential(api, frames, params, guesses)
    home = api.Context(0)
    dev = {k: api.DeviceCloud(f, home) for k, f in enumerate(frames)}
    abi_guesses = {k: api._colmajor(g) for k, g in guesses.items()} if guesses else None
    for in_flight in (1, 2, 3, 4):
        reg = chain.ChainRegistrar(0, in_flight, params=params)
        seen = {}

        def collect(k, r, c):
            seen[k] = (r.iterations, bool(r.converged), int(r.n_correspondences))

        for _rep in range(2):      # the second round meets warm contexts with carried tile schedules
            got = reg.register(dev, range(1, N_FRAMES), abi_guesses, collect=collect)
            assert sorted(got) == list(range(1, N_FRAMES))
            for k in range(1, N_FRAMES):
                np.testing.assert_array_equal(got[k], alone[k][0], err_msg="pair %d, %d in flight" % (k, in_flight))
                assert seen[k] == alone[k][1:], (k, in_flight)
        # (which context got which pair is a matter of timing -- a late thread may find the queue empty -- and is not asserted)
        assert set(reg.pair_context.values()) <= set(range(in_flight))


def test_pairs_in_flight_match_the_oracle(env, frames, orc):
    api, chain, lib = env
    params = api.icp_params(reference=True)
    home = api.Context(0)
    dev = {k: api.DeviceCloud(f, home) for k, f in enumerate(frames[:3])}
    got = chain.ChainRegistrar(0, 2, params=params).register(dev, [1, 2])
    for k in (1, 2):
        o = orc.IcpOracle()
        o.set_target(frames[k - 1].points)
        o.set_source(frames[k].points)
        r = o.align(None, orc.IcpParams.reference())
        assert np.linalg.norm(got[k] - r.T) < 1e-5        # tolerance of the parity suite (north star: 1e-4 Frobenius)
    poses = chain.compose_chain(got, 3)
    np.testing.assert_allclose(poses[2], np.asarray(got[1], np.float64) @ np.asarray(got[2], np.float64))


def test_a_failing_pair_is_reported_not_swallowed(env, frames):
    api, chain, lib = env
    home = api.Context(0)
    dev = {0: api.DeviceCloud(frames[0], home), 1: api.DeviceCloud(frames[1], home)}
    bad = api.icp_params(reference=True)
    bad.max_iterations = -1
    with pytest.raises(lib.RsregError):
        chain.ChainRegistrar(0, 2, params=bad).register(dev, [1])


@pytest.fixture(scope="module")
def runner():
    out = os.path.join(ROOT, "tests", "cpp", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "scheme_runner_chain")
    pkg = os.path.join(ROOT, "realsense-pointcloud_amd")
    cmd = ["g++", "-std=c++17", "-O2", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "scheme_runner.cpp"),
           "-o", exe, "-L", pkg, "-lrsreg", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    return exe


def _read_chain(path, n):
    rows = [l.split() for l in open(path).read().strip().splitlines()]
    heads, pairs, at = [], [], 0
    for _k in range(1, n):
        heads.append([int(v) for v in rows[at]])
        pairs.append(np.array(rows[at + 1:at + 5], dtype=np.float64))
        at += 5
    poses = np.array(rows[at:], dtype=np.float64).reshape(n, 4, 4)
    return heads, np.array(pairs), poses


@pytest.mark.parametrize("iters", ["0", "12"])
def test_cpp_chain_registrar(env, frames, runner, tmp_path, rs, iters):
    """rsreg::ChainRegistrar from host frames: K = 1 and K = 3 give the same bits as each other and as the Python path."""
    api, chain, lib = env
    paths = []
    for k, f in enumerate(frames):
        p = str(tmp_path / ("f%d.pcd" % k))
        rs.cloud.save_pcd(p, f)
        paths.append(p)
    res = {}
    for k_in_flight in ("1", "3"):
        pre = str(tmp_path / ("chain_" + k_in_flight))
        subprocess.run([runner, "chain", pre] + paths, check=True,
                       env=dict(os.environ, RSREG_CHAIN_IN_FLIGHT=k_in_flight, RSREG_CHAIN_ITERATIONS=iters, RSREG_SCHEME_TIME="2"))
        res[k_in_flight] = _read_chain(pre + ".txt", N_FRAMES)
    h1, p1, poses1 = res["1"]
    h3, p3, poses3 = res["3"]
    np.testing.assert_array_equal(p1, p3)
    np.testing.assert_array_equal(poses1, poses3)
    assert [h[:2] for h in h1] == [h[:2] for h in h3]
    assert {h[2] for h in h1} == {0} and {h[2] for h in h3} <= {0, 1, 2}
    if iters == "0":
        params = api.icp_params(reference=True)
    else:
        params = api.icp_params(max_iterations=12, criteria_mode=1, pipeline_mode=2, max_correspondence_distance=0.05)
    alone = _sequential(api, frames, params, None)
    for k in range(1, N_FRAMES):
        # (%.9g round-trips a float32)
        np.testing.assert_array_equal(p3[k - 1].astype(np.float32), alone[k][0])
        assert h3[k - 1][0] == int(alone[k][2]) and h3[k - 1][1] == alone[k][1]
    want = chain.compose_chain({k: alone[k][0] for k in alone}, N_FRAMES)
    np.testing.assert_allclose(poses3, np.array(want), atol=2e-7)


def test_bench_chain_in_flight():
    """`bench.py --workload chain --frames F --in-flight K`: the contract line, config.in_flight, and the bench's own
    cross-check of every pair against the sequential cloud-handle path."""
    r = subprocess.run([sys.executable, "bench.py", "--workload", "chain", "--size", "50k", "--frames", "6", "--in-flight", "3", "--steps", "2",
                        "--warmup", "1", "--cpu-iterations", "2"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["config"]["in_flight"] == 3 and j["config"]["n_frames"] == 6 and j["config"]["pairs"] == 5 and j["pairs_on_rank0"] == 5
    assert j["in_flight_vs_sequential_max_abs_diff"] == 0.0
    assert j["value"] > 0 and j["roofline"]["achieved"] > 0 and j["cpu_baseline"]["value"] > 0
    assert j["transform_error_vs_cpu_frobenius"] < 1e-4 and j["chain_pose_error_vs_ground_truth_frobenius_max"] < 0.1
