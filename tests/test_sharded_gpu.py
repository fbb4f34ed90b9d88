"""GPU: the N-rank path with the HIP engine.  (1) two processes share the one GPU of the box,
each holds one source block, the 17 sums travel over torch.distributed/gloo; the result must
equal the single-process engine's.  (2) the native RCCL transport (rsreg_comm_*) on a one-rank
communicator: init, all-reduce inside rsreg_icp_align, destroy."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import rsreg_amd
    from rsreg_amd import api, sharded, synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tgt = synth.render_frame(0, "50k", "bench")
    src = synth.render_frame(1, "50k", "bench")
    guess = synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
    lo, hi = sharded.shard_range(len(src), rank, world)

    def allreduce(v):
        t = torch.from_numpy(v.copy())
        dist.all_reduce(t)
        return t.numpy()

    # one device per rank where the box has them (the first multi-GPU box then runs this test across GPUs by itself)
    icp = api.IterativeClosestPoint(api.Context(rank if api.device_count() >= world else 0))
    icp.params = api.icp_params(max_iterations=6, criteria_mode=1, max_correspondence_distance=0.05)
    icp.setInputSource(np.ascontiguousarray(src.points[lo:hi]))
    icp.setInputTarget(tgt)
    r = sharded.run_sharded_icp(icp, allreduce, guess)
    np.save(os.path.join(out_dir, "T_rank%d.npy" % rank), api._rowmajor(r.transform))
    np.save(os.path.join(out_dir, "meta_rank%d.npy" % rank), np.array([r.iterations, r.state, r.converged, r.n_correspondences]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_matches_single_process(tmp_path, rs):
    import torch.multiprocessing as mp
    from rsreg_amd import api, lib, synth
    lib.build()
    if api.device_count() < 1:
        pytest.fail("no HIP device: the product has no CPU fallback")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    tgt = synth.render_frame(0, "50k", "bench")
    src = synth.render_frame(1, "50k", "bench")
    guess = synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
    icp = api.IterativeClosestPoint()
    icp.params = api.icp_params(max_iterations=6, criteria_mode=1, max_correspondence_distance=0.05)
    icp.setInputSource(src)
    icp.setInputTarget(tgt)
    icp.align(guess)
    T0, T1 = np.load(tmp_path / "T_rank0.npy"), np.load(tmp_path / "T_rank1.npy")
    np.testing.assert_array_equal(T0, T1)
    assert np.linalg.norm(T0 - icp.getFinalTransformation()) < 1e-6
    m0, m1 = np.load(tmp_path / "meta_rank0.npy"), np.load(tmp_path / "meta_rank1.npy")
    np.testing.assert_array_equal(m0, m1)
    assert tuple(m0) == (icp.result.iterations, icp.result.state, icp.result.converged, icp.result.n_correspondences)


def _worker_native(rank, world, port, out_dir):
    """rank r on device r: ncclCommInitRank(nranks = world) and the all-reduce of the 17 sums inside rsreg_icp_align"""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import rsreg_amd  # noqa: F401
    from rsreg_amd import api, sharded, synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)       # (only to hand the 128-byte id around)
    box = [api.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    ctx = api.Context(rank)
    ctx.comm_init(box[0], rank, world)
    tgt = synth.render_frame(0, "50k", "bench")
    src = synth.render_frame(1, "50k", "bench")
    guess = synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
    lo, hi = sharded.shard_range(len(src), rank, world)
    out = []
    for pipeline in (1, 2):                                            # host solve per iteration; device-resident loop
        icp = api.IterativeClosestPoint(ctx)
        icp.params = api.icp_params(max_iterations=6, criteria_mode=1, pipeline_mode=pipeline, max_correspondence_distance=0.05)
        icp.setInputSource(np.ascontiguousarray(src.points[lo:hi]))
        icp.setInputTarget(tgt)
        icp.align(guess)
        out.append(api._rowmajor(icp.result.transform))
        out.append(np.array([icp.result.iterations, icp.result.state, icp.result.converged, icp.result.n_correspondences], np.float64).reshape(1, 4).repeat(4, 0))
    np.save(os.path.join(out_dir, "native_rank%d.npy" % rank), np.stack(out))
    dist.barrier()
    dist.destroy_process_group()


def test_native_rccl_transport_two_ranks_two_gpus(tmp_path, rs):
    """BASELINE configs[3] over the native transport with more than one rank: needs two GPUs (RCCL refuses two ranks on
    one device), so it runs by itself on the first multi-GPU box and is skipped, visibly, on the one-GPU boxes."""
    import torch.multiprocessing as mp
    from rsreg_amd import api, lib, synth
    lib.build()
    if api.device_count() < 2:
        pytest.skip("one GPU on this box: ncclCommInitRank(nranks = 2) needs two devices (the 1-rank communicator is tested below)")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_native, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "native_rank0.npy"), np.load(tmp_path / "native_rank1.npy")
    np.testing.assert_array_equal(a, b)                                # every rank solves the same sums: identical bits
    np.testing.assert_array_equal(a[0], a[2])                          # both pipelines
    tgt, src = synth.render_frame(0, "50k", "bench"), synth.render_frame(1, "50k", "bench")
    icp = api.IterativeClosestPoint()
    icp.params = api.icp_params(max_iterations=6, criteria_mode=1, max_correspondence_distance=0.05)
    icp.setInputSource(src)
    icp.setInputTarget(tgt)
    icp.align(synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32))
    assert np.linalg.norm(a[0] - icp.getFinalTransformation()) < 1e-6
    assert tuple(a[1][0]) == (icp.result.iterations, icp.result.state, icp.result.converged, icp.result.n_correspondences)


def test_native_rccl_transport_single_rank(rs):
    from rsreg_amd import api, lib, synth
    lib.build()
    ctx = api.Context(0)
    ctx.comm_init(api.comm_unique_id(), 0, 1)
    v = ctx.allreduce_f64(np.arange(17, dtype=np.float64))
    np.testing.assert_array_equal(v, np.arange(17))
    tgt = synth.render_frame(0, "50k", "parity")
    src = synth.render_frame(1, "50k", "parity")
    a = api.IterativeClosestPoint(ctx)
    a.params = api.icp_params(reference=True)
    a.setInputSource(src)
    a.setInputTarget(tgt)
    a.align()
    b = api.IterativeClosestPoint(api.Context(0))
    b.params = api.icp_params(reference=True)
    b.setInputSource(src)
    b.setInputTarget(tgt)
    b.align()
    np.testing.assert_array_equal(a.getFinalTransformation(), b.getFinalTransformation())
    # the device-resident loop with the all-reduce between its reduce and solve kernels
    out = []
    for icp in (a, b):
        for pipeline in (1, 2):
            icp.params = api.icp_params(max_iterations=5, criteria_mode=1, pipeline_mode=pipeline, max_correspondence_distance=0.01)
            icp.align()
            out.append((bytes(icp.result.transform), bytes(icp.result.sums_last), icp.result.iterations, icp.result.state))
    assert out[0] == out[1] == out[2] == out[3]
    assert lib.lib().rsreg_comm_destroy(ctx.h) == 0


def test_ndt_shards_by_source_blocks(rs):
    """NDT shards like ICP: the score / gradient / Hessian of source blocks add up to the whole cloud's (what N
    ranks all-reduce: 28 doubles per pass), and with a communicator attached rsreg_ndt_align runs that all-reduce
    on the stream in every pass (here over one rank: RCCL refuses two ranks on the one device of the box)."""
    from rsreg_amd import api, lib, synth
    lib.build()
    tgt = synth.render_frame(0, "50k", "bench")
    src = synth.render_frame(1, "50k", "bench")
    pose = np.array([0.01, -0.005, 0.008, 0.002, 0.03, -0.001])
    whole = api.NormalDistributionsTransform(api.Context(0))
    whole.params = api.ndt_params(reference=True)
    whole.setInputSource(src)
    whole.setInputTarget(tgt)
    s_all, g_all, h_all = whole.derivatives(pose)
    n = len(src)
    acc = [0.0, np.zeros(6), np.zeros((6, 6))]
    for lo, hi in ((0, n // 3), (n // 3, n // 2), (n // 2, n)):
        part = api.NormalDistributionsTransform(api.Context(0))
        part.params = api.ndt_params(reference=True)
        part.setInputSource(np.ascontiguousarray(src.points[lo:hi]))
        part.setInputTarget(tgt)
        s, g, h = part.derivatives(pose)
        acc[0] += s
        acc[1] += g
        acc[2] += h
    assert abs(acc[0] - s_all) < 1e-9 * abs(s_all)
    np.testing.assert_allclose(acc[1], g_all, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(acc[2], h_all, rtol=1e-9, atol=1e-7)
    # the collective inside rsreg_ndt_align (one-rank communicator): same result as without one
    ctx = api.Context(0)
    ctx.comm_init(api.comm_unique_id(), 0, 1)
    a = api.NormalDistributionsTransform(ctx)
    a.params = api.ndt_params(reference=True)
    a.setInputSource(src)
    a.setInputTarget(tgt)
    a.align(synth.small_transform(1.0, (0, 0, 0)).astype(np.float32))
    whole.align(synth.small_transform(1.0, (0, 0, 0)).astype(np.float32))
    np.testing.assert_array_equal(a.getFinalTransformation(), whole.getFinalTransformation())
    assert (a.result.iterations, a.result.n_derivative_passes) == (whole.result.iterations, whole.result.n_derivative_passes)
    assert lib.lib().rsreg_comm_destroy(ctx.h) == 0
