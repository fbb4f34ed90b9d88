"""One cloud pair registered by N ranks: the source is cut into blocks dealt to the ranks, the target
(and its index) is replicated, and the only exchange per iteration is the sum of the 17
per-block sums (BASELINE configs[3]).  Every rank then runs the same host
Umeyama/SVD on identical numbers, so all ranks hold the same transform without a broadcast.

The reference has no distributed path; this module is new work.  Two transports:
  * native: ``Context.comm_init`` + ``rsreg_icp_align`` all-reduces on the device over RCCL
    (what bench.py uses on a multi-GPU node);
  * step-wise: ``run_sharded_icp`` below drives begin/search/sums/update and leaves the
    all-reduce to a callable (torch.distributed with gloo or nccl, or RCCL via the ctx).
"""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous block [lo, hi) of rank `rank` when n items are cut into `world` blocks."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return (n * rank) // world, (n * (rank + 1)) // world


def shard_blocks(n, rank, world, block=256):
    """Indices of rank `rank` when n items are cut into blocks of `block` consecutive items and
    the blocks are dealt round-robin.  Same exchange as `shard_range` (the sums do not care which
    points a rank holds), but every rank gets an even sample of the scene: the slow waves of a
    launch (queries on near, densely sampled surfaces) are spread over all ranks instead of
    landing on the one that owns that part of the image."""
    if world < 1 or not (0 <= rank < world) or block < 1:
        raise ValueError("bad rank/world/block")
    idx = np.arange(n, dtype=np.int64)
    return idx[(idx // block) % world == rank]


def run_sharded_icp(stepper, allreduce, guess=None, max_steps=100000):
    """Drive one ICP alignment whose sums are combined across ranks.

    stepper   : an object with begin(guess) / search() / sums() / update(sums) / end() — the
                step-wise form of ``api.IterativeClosestPoint`` holding THIS rank's source block.
    allreduce : callable(np.ndarray[17] float64) -> the element-wise sum over all ranks.
    Returns the result of stepper.end().  All ranks take the same number of iterations because
    they evaluate the convergence criteria on the same (global) sums.
    """
    stepper.begin(guess)
    for _ in range(max_steps):
        stepper.search(False) if getattr(stepper, "_quiet_search", False) else stepper.search()
        local = np.ascontiguousarray(stepper.sums(), np.float64)
        total = np.ascontiguousarray(allreduce(local), np.float64)
        _, done = stepper.update(total)
        if done:
            break
    return stepper.end()
