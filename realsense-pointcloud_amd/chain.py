"""A chain of frames registered as independent consecutive pairs, one pair per GPU (BASELINE
configs[4]).

The reference's IncrementalICP (src/incremental_icp.hpp:51-66) is sequential by definition:
frame k is registered against the model accumulated from the frames before it.  The throughput
configuration restates the chain as the consecutive pairs (k-1, k) -- independent of one another,
so they shard over ranks with no exchange on the data path -- and composes the pair transforms on
the host afterwards: T_0k = T_01 * T_12 * ... * T_(k-1)k.  That is a documented deviation from
frame-to-model registration; the pair registrations themselves are the same ICP.
"""
import ctypes as C
import threading

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


class ChainRegistrar:
    """Pairs (k-1, k) of frames resident in HBM, `in_flight` of them side by side on ONE GPU (the Python mirror of
    rsreg::ChainRegistrar, include/rsreg/schemes.hpp).

    One alignment leaves a third of the chip-time of every search launch to an emptying tail and a few microseconds of
    nothing between two dependent launches; kernels of other alignments fill both.  Every pair in flight
    runs on a context of its own (stream, scratch, index; a context is not thread-safe, different contexts are
    independent: include/rsreg.h) driven by a thread of its own, and reads the frames where their owner has put them
    (rsreg_cloud_device_ptr: a settled cloud may be read by any stream).  A pair gets the bits it gets alone.
    The pairs are the same ICP as the reference's (src/incremental_icp.hpp:57-59); registering consecutive pairs
    instead of frame-to-model is the documented restatement of configs[4] (module docstring)."""

    def __init__(self, device=0, in_flight=3, params=None, contexts=None):
        from . import api, lib
        self._api, self._lib = api, lib
        self.device = device
        self.params = params if params is not None else api.icp_params(reference=True)
        self.contexts = list(contexts) if contexts else []
        self.in_flight = 0
        self.set_in_flight(in_flight)
        self.pair_context = {}

    def set_in_flight(self, k):
        k = max(1, int(k))
        while len(self.contexts) < k:      # contexts are kept (with their buffers) from one call to the next
            self.contexts.append(self._api.Context(self.device))
        self.in_flight = k

    @staticmethod
    def share(cloud):
        """(address, n, stride, is_dense) of a DeviceCloud's records, settled: complete in HBM, readable from any stream."""
        n, stride, _w, _h, dense = cloud.info()
        return (cloud.device_ptr, n, stride, dense)

    def register(self, frames, ks, guesses=None, collect=None):
        """frames: {k: DeviceCloud or share() tuple}; ks: the pairs (k-1, k) to register; guesses: {k: 4x4 column-major float32
        as the ABI takes it} or None (identity, incremental_icp.hpp:59).  Returns {k: 4x4 numpy, frame k into frame k-1}.
        collect (optional): callable(k, IcpResult, ctx) run on the worker's thread after every pair."""
        L, lib, api = self._lib.lib(), self._lib, self._api
        shared = {k: (f if isinstance(f, tuple) else self.share(f)) for k, f in frames.items()}
        ks = list(ks)
        out, errors = {}, []
        it = iter(ks)
        lock = threading.Lock()
        prm = self.params

        def work(w):
            ctx = self.contexts[w]
            res = lib.IcpResult()
            try:
                while True:
                    with lock:
                        k = next(it, None)
                    if k is None or errors:
                        return
                    sp, sn, ss, sd = shared[k]
                    tp, tn, ts, td = shared[k - 1]
                    g = guesses[k] if guesses is not None else None
                    # the source first, the reference's order: it is put into the engine's order beside the target's index build
                    lib.check(L.rsreg_icp_set_source_device(ctx.h, C.c_void_p(sp), sn, ss, int(sd)), ctx.h)
                    lib.check(L.rsreg_icp_set_target_device(ctx.h, C.c_void_p(tp), tn, ts, int(td), prm.max_correspondence_distance), ctx.h)
                    lib.check(L.rsreg_icp_align(ctx.h, g.ctypes.data if g is not None else None, C.byref(prm), C.byref(res), None, 0), ctx.h)
                    T = api._rowmajor(res.transform)
                    if collect is not None:
                        collect(k, res, ctx)
                    with lock:
                        out[k] = T
                        self.pair_context[k] = w
            except Exception as e:   # noqa: BLE001 -- handed to the caller's thread
                errors.append(e)

        n_workers = min(self.in_flight, max(1, len(ks)))
        threads = [threading.Thread(target=work, args=(w,)) for w in range(1, n_workers)]
        for t in threads:
            t.start()
        work(0)    # the caller's thread drives a context too
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return out
