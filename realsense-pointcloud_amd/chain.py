"""A chain of frames registered as independent consecutive pairs, one pair per GPU (BASELINE
configs[4]).

The reference's IncrementalICP (src/incremental_icp.hpp:51-66) is sequential by definition:
frame k is registered against the model accumulated from the frames before it.  The throughput
configuration restates the chain as the consecutive pairs (k-1, k) -- independent of one another,
so they shard over ranks with no exchange on the data path -- and composes the pair transforms on
the host afterwards: T_0k = T_01 * T_12 * ... * T_(k-1)k.  That is a documented deviation from
frame-to-model registration; the pair registrations themselves are the same ICP.
"""
import numpy as np


def pair_assignment(n_frames, rank, world):
    """Pairs (k-1, k), k = 1 .. n_frames-1, dealt round-robin: the k of the pairs rank `rank` registers."""
    if world < 1 or not (0 <= rank < world) or n_frames < 1:
        raise ValueError("bad rank/world/n_frames")
    return [k for k in range(1, n_frames) if (k - 1) % world == rank]


def gather_pairs(local, n_frames, allgather):
    """local: {k: 4x4} of this rank.  allgather: callable(np.ndarray[(n_frames, 17) float64]) ->
    list of every rank's array (a fixed-shape exchange, so any backend's all_gather will do).
    Returns {k: 4x4} of ALL pairs; raises if a pair is missing or was registered twice."""
    buf = np.zeros((n_frames, 17), np.float64)
    for k, T in local.items():
        buf[k, 0] = 1.0
        buf[k, 1:] = np.asarray(T, np.float64).reshape(16)
    allbufs = allgather(buf)
    seen = np.zeros(n_frames)
    out = {}
    for b in allbufs:
        b = np.asarray(b, np.float64).reshape(n_frames, 17)
        for k in np.nonzero(b[:, 0])[0]:
            seen[k] += 1
            out[int(k)] = b[k, 1:].reshape(4, 4)
    if (seen[1:] != 1).any() or seen[0] != 0:
        raise RuntimeError("pairs registered %s times" % seen[1:].tolist())
    return out


def compose_chain(pairs, n_frames):
    """pairs[k] maps frame k into frame k-1; returns [T_00 = I, T_01, ..., T_0(n-1)]: frame k into frame 0."""
    out = [np.eye(4)]
    for k in range(1, n_frames):
        out.append(out[-1] @ np.asarray(pairs[k], np.float64))
    return out
