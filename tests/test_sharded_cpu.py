"""CPU, 2 processes over gloo: the N-rank protocol of realsense-pointcloud_amd/sharded.py —
source cut into contiguous blocks, 17 sums all-reduced per iteration, identical host solve on
every rank.  On this CPU box the per-block sums come from the oracle (the device kernels need a
GPU; tests/test_sharded_gpu.py repeats this with the HIP engine), the all-reduce is real
torch.distributed/gloo, and the Umeyama step is the product's host code (rsreg_umeyama_from_sums).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleStepper:
    """The oracle behind the stepper interface run_sharded_icp expects; the transform of each
    iteration is computed by the PRODUCT's host Umeyama from the all-reduced sums."""

    def __init__(self, src_block, tgt, params):
        import oracle
        from rsreg_amd import api
        self.api = api
        self.o = oracle.IcpOracle()
        self.o.set_target(tgt)
        self.o.set_source(src_block)
        self.params = params

    def begin(self, guess):
        self.o.begin(guess, self.params)

    def search(self):
        return self.o.search()

    def sums(self):
        return self.o.sums()

    def update(self, sums):
        t_inc, done = self.o.update(sums)
        if sums[0] >= 3:
            # product host code must give the same increment from the same global sums
            np.testing.assert_allclose(self.api.umeyama_from_sums(sums), t_inc, atol=1e-7)
        return t_inc, done

    def end(self):
        return self.o.end()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import oracle
    import rsreg_amd
    from rsreg_amd import sharded, synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tgt = synth.render_frame(0, "50k", "bench")
    src = synth.render_frame(1, "50k", "bench")
    guess = synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
    lo, hi = sharded.shard_range(len(src), rank, world)
    p = oracle.IcpParams.default()
    p.max_iterations, p.criteria_mode, p.max_correspondence_distance = 6, 1, 0.05

    def allreduce(v):
        t = torch.from_numpy(v.copy())
        dist.all_reduce(t)
        return t.numpy()

    st = OracleStepper(np.ascontiguousarray(src.points[lo:hi]), tgt.points, p)
    r = sharded.run_sharded_icp(st, allreduce, guess)
    np.save(os.path.join(out_dir, "T_rank%d.npy" % rank), r.T)
    np.save(os.path.join(out_dir, "meta_rank%d.npy" % rank), np.array([r.iterations, r.state, r.converged, r.n_correspondences]))
    if rank == 0:
        o = oracle.IcpOracle()
        o.set_target(tgt.points)
        o.set_source(src.points)
        ref = o.align(guess, p)
        np.save(os.path.join(out_dir, "T_ref.npy"), ref.T)
        np.save(os.path.join(out_dir, "meta_ref.npy"), np.array([ref.iterations, ref.state, ref.converged, ref.n_correspondences]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly(rs):
    from rsreg_amd import sharded
    for n in (0, 1, 7, 1000, 1000000, 307200):
        for world in (1, 2, 3, 8):
            spans = [sharded.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharded.shard_range(10, 2, 2)


def test_shard_blocks_partition_exactly(rs):
    from rsreg_amd import sharded
    for n in (0, 1, 7, 1000, 307200, 1000000):
        for world in (1, 2, 3, 8):
            for block in (1, 64, 256):
                parts = [sharded.shard_blocks(n, r, world, block) for r in range(world)]
                allidx = np.sort(np.concatenate(parts)) if n else np.zeros(0, np.int64)
                assert np.array_equal(allidx, np.arange(n))
                if n >= world * block * 8:
                    sizes = [len(p) for p in parts]
                    assert max(sizes) - min(sizes) <= block
    assert np.array_equal(sharded.shard_blocks(10, 0, 1), np.arange(10))
    with pytest.raises(ValueError):
        sharded.shard_blocks(10, 2, 2)


def test_two_rank_icp_over_gloo_matches_single_process(tmp_path, orc, rs):
    import torch.multiprocessing as mp
    from rsreg_amd import lib
    lib.build()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    T0, T1, Tref = (np.load(tmp_path / f) for f in ("T_rank0.npy", "T_rank1.npy", "T_ref.npy"))
    np.testing.assert_array_equal(T0, T1)                 # every rank holds the same transform, bit for bit
    assert np.linalg.norm(T0 - Tref) < 1e-6               # and it is the single-process answer
    m0, m1, mref = (np.load(tmp_path / f) for f in ("meta_rank0.npy", "meta_rank1.npy", "meta_ref.npy"))
    np.testing.assert_array_equal(m0, m1)
    np.testing.assert_array_equal(m0, mref)
