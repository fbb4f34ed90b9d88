"""CPU: the chain-as-consecutive-pairs workload (BASELINE configs[4], rsreg_amd/chain.py) over two
gloo ranks: pair assignment, the fixed-shape gather of the 4x4s and the composition on the host.
The per-pair engine here is the CPU oracle (TEST INFRASTRUCTURE; on a GPU the same logic runs in
bench.py --workload chain with the HIP engine)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pair_assignment_partitions_exactly(rs):
    from rsreg_amd import chain
    for n_frames in (1, 2, 5, 16, 17):
        for world in (1, 2, 3, 8):
            got = sorted(k for r in range(world) for k in chain.pair_assignment(n_frames, r, world))
            assert got == list(range(1, n_frames))
            sizes = [len(chain.pair_assignment(n_frames, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert chain.pair_assignment(16, 0, 8) == [1, 9] and chain.pair_assignment(16, 7, 8) == [8]
    with pytest.raises(ValueError):
        chain.pair_assignment(4, 2, 2)


def test_compose_and_gather_checks(rs):
    from rsreg_amd import chain, synth
    pairs = {k: synth.ground_truth(k, k - 1, "bench") for k in range(1, 6)}
    poses = chain.compose_chain(pairs, 6)
    for k in range(6):
        np.testing.assert_allclose(poses[k], synth.ground_truth(k, 0, "bench"), atol=1e-12)
    one = chain.gather_pairs(pairs, 6, lambda b: [b])
    assert sorted(one) == [1, 2, 3, 4, 5]
    with pytest.raises(RuntimeError):
        chain.gather_pairs({1: np.eye(4)}, 3, lambda b: [b])          # pair 2 missing
    with pytest.raises(RuntimeError):
        chain.gather_pairs({1: np.eye(4), 2: np.eye(4)}, 3, lambda b: [b, b])   # registered twice


def _worker(rank, world, port, out_dir, n_frames):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import oracle
    import rsreg_amd  # noqa: F401
    from rsreg_amd import chain, synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = chain.pair_assignment(n_frames, rank, world)
    local = {}
    p = oracle.IcpParams.default()
    p.max_iterations, p.criteria_mode, p.max_correspondence_distance = 8, 1, 0.05
    for k in mine:
        tgt, src = synth.render_frame(k - 1, (80, 60), "bench"), synth.render_frame(k, (80, 60), "bench")
        o = oracle.IcpOracle()
        o.set_target(tgt.points, dedup=True)
        o.set_source(src.points)
        local[k] = o.align(synth.ground_truth(k, k - 1, "bench").astype(np.float32), p).T

    def allgather(buf):
        t = torch.from_numpy(buf.copy())
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return [o.numpy() for o in outs]

    poses = chain.compose_chain(chain.gather_pairs(local, n_frames, allgather), n_frames)
    np.save(os.path.join(out_dir, "poses_rank%d.npy" % rank), np.stack(poses))
    np.save(os.path.join(out_dir, "mine_rank%d.npy" % rank), np.array(mine))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_chain_over_gloo(tmp_path, orc, rs):
    import torch.multiprocessing as mp
    from rsreg_amd import chain, synth
    n_frames = 5
    port = 29700 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path), n_frames), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "poses_rank0.npy"), np.load(tmp_path / "poses_rank1.npy")
    np.testing.assert_array_equal(p0, p1)                      # every rank composes the same chain
    assert sorted(np.load(tmp_path / "mine_rank0.npy").tolist() + np.load(tmp_path / "mine_rank1.npy").tolist()) == [1, 2, 3, 4]
    # one process doing all pairs gives the same chain
    pr = orc.IcpParams.default()
    pr.max_iterations, pr.criteria_mode, pr.max_correspondence_distance = 8, 1, 0.05
    pairs = {}
    for k in range(1, n_frames):
        tgt, src = synth.render_frame(k - 1, (80, 60), "bench"), synth.render_frame(k, (80, 60), "bench")
        o = orc.IcpOracle()
        o.set_target(tgt.points, dedup=True)
        o.set_source(src.points)
        pairs[k] = o.align(synth.ground_truth(k, k - 1, "bench").astype(np.float32), pr).T
    np.testing.assert_array_equal(p0, np.stack(chain.compose_chain(pairs, n_frames)))
    for k in range(n_frames):                                   # and it stays near the true poses
        assert np.linalg.norm(p0[k] - synth.ground_truth(k, 0, "bench")) < 0.03 * (k + 1)   # (80 x 60-pixel frames: coarse)
