"""The tile schedule of the fused search kernel (csrc/icp.hip: build_schedule): tiles launched longest first,
the longest ones searched by 2 or 4 lanes per query.  It may change WHEN a query is searched and by how many
lanes, never what is found or how the 17 sums are added up: every result must be bit-identical to the
unscheduled launch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(rs):
    from rsreg_amd import api as a, lib
    lib.build()
    if a.device_count() < 1:
        pytest.fail("no HIP device: the product has no CPU fallback")
    return a


@pytest.fixture(scope="module")
def frames(rs):
    return {("50k", "parity"): (rs.synth.render_frame(1, "50k", "parity"), rs.synth.render_frame(0, "50k", "parity"))}


LAST = {"scheduled": 0}   # rsreg_icp_result.n_scheduled_launches of the last _run


def _run(api, src, tgt, pipeline, iters, gate, guess=None):
    icp = api.IterativeClosestPoint(api.Context(0))
    icp.params = api.icp_params(max_iterations=iters, criteria_mode=1, pipeline_mode=pipeline, max_correspondence_distance=gate)
    icp.setInputSource(src)
    icp.setInputTarget(tgt)
    out = icp.align(guess) if guess is not None else icp.align()
    r = icp.result
    LAST["scheduled"] = int(r.n_scheduled_launches)
    kind = icp.grid_info().index_kind
    return (bytes(r.transform), bytes(r.sums_last), r.n_correspondences, r.iterations, r.state, r.converged, r.mse,
            np.stack([out.points[k] for k in "xyz"]).tobytes()), kind


@pytest.mark.parametrize("f4,f2,at", [(0.0, 0.10, 1), (0.25, 0.25, 1), (0.25, 0.0, 0), (0.0, 0.25, 2), (0.02, 0.08, 1)])
def test_scheduled_launches_change_nothing(api, frames, monkeypatch, capfd, f4, f2, at):
    src, tgt = frames[("50k", "parity")]
    monkeypatch.setenv("RSREG_SCHED", "0")
    base, kind = _run(api, src, tgt, 2, 8, 0.02)
    if kind != 1:
        pytest.skip("the schedule belongs to the dense-table search")
    monkeypatch.setenv("RSREG_SCHED", "1")
    monkeypatch.setenv("RSREG_SCHED_MIN_TILES", "1")
    monkeypatch.setenv("RSREG_SCHED_F4", str(f4))
    monkeypatch.setenv("RSREG_SCHED_F2", str(f2))
    monkeypatch.setenv("RSREG_SCHED_AT", str(at))
    capfd.readouterr()
    for pipeline in (1, 2):
        got, _ = _run(api, src, tgt, pipeline, 8, 0.02)
        assert LAST["scheduled"] > 0, "the schedule was not built"
        assert got == base, (pipeline, f4, f2)


def test_schedule_on_ragged_and_invalid_input(api, rs, monkeypatch, capfd):
    """A source whose size is not a multiple of the tile, with invalid and duplicate records; every tile split."""
    rng = np.random.default_rng(5)
    tgt = rs.synth.render_frame(0, "50k", "parity")
    src = rs.synth.render_frame(1, "50k", "parity")
    pts = src.points.copy()
    n = (len(pts) // 128) * 128 - 37
    pts = pts[:n]
    bad = rng.integers(0, n, n // 40)
    pts["x"][bad] = np.nan
    pts[rng.integers(0, n, n // 30)] = pts[0]
    src = rs.PointCloud(pts, width=n, height=1, is_dense=False)
    monkeypatch.setenv("RSREG_SCHED", "0")
    base, kind = _run(api, src, tgt, 2, 6, 0.03)
    if kind != 1:
        pytest.skip("the schedule belongs to the dense-table search")
    monkeypatch.setenv("RSREG_SCHED", "1")
    monkeypatch.setenv("RSREG_SCHED_MIN_TILES", "1")
    monkeypatch.setenv("RSREG_SCHED_F4", "0.25")
    monkeypatch.setenv("RSREG_SCHED_F2", "0.25")
    capfd.readouterr()
    got, _ = _run(api, src, tgt, 2, 6, 0.03)
    assert LAST["scheduled"] > 0
    assert got == base


def test_schedule_at_bench_size(api, rs, monkeypatch):
    """The default settings on a 300 k pair (the schedule switches itself on from 1024 tiles)."""
    tgt, src = rs.synth.render_frame(0, "N300", "bench"), rs.synth.render_frame(1, "N300", "bench")
    guess = rs.synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
    monkeypatch.setenv("RSREG_SCHED", "0")
    base, _ = _run(api, src, tgt, 2, 12, 0.05, guess)
    monkeypatch.delenv("RSREG_SCHED")
    for _ in range(3):   # (the parts of a split tile meet through device-scope atomics: repeat)
        got, _ = _run(api, src, tgt, 2, 12, 0.05, guess)
        assert got == base


@pytest.mark.parametrize("f4,f2", [(0.0, 1.0), (1.0, 0.0), (0.3, 0.4)])
def test_every_tile_split_at_1m(api, rs, monkeypatch, f4, f2):
    """All 6985 tiles of the 1 M bench pair searched by 2 or 4 lanes per query (crowded cells, rows beyond ring 1
    whose extents differ from lane to lane): same bits as the unscheduled launch.  (The lanes of a split query
    each clip an outer row to what their OWN best still allows: chunk ownership has to be by absolute position.)"""
    tgt, src = rs.synth.render_frame(0, "N1M", "bench"), rs.synth.render_frame(1, "N1M", "bench")
    guess = rs.synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
    monkeypatch.setenv("RSREG_SCHED", "0")
    base, kind = _run(api, src, tgt, 2, 5, 0.05, guess)
    if kind != 1:
        pytest.skip("the schedule belongs to the dense-table search")
    monkeypatch.setenv("RSREG_SCHED", "1")
    monkeypatch.setenv("RSREG_SCHED_MIN_TILES", "1")
    monkeypatch.setenv("RSREG_SCHED_F4", str(f4))
    monkeypatch.setenv("RSREG_SCHED_F2", str(f2))
    got, _ = _run(api, src, tgt, 2, 5, 0.05, guess)
    assert got == base


def test_contexts_in_flight_on_one_gpu(api, rs, monkeypatch):
    """Three alignments at once (one context and one host thread each) share the GPU: the parts of a split tile then
    run far apart in time, and every context must still get the bits it gets alone."""
    import threading
    tgt, src = rs.synth.render_frame(0, "N1M", "bench"), rs.synth.render_frame(1, "N1M", "bench")
    guess = rs.synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
    monkeypatch.delenv("RSREG_SCHED", raising=False)
    alone, _ = _run(api, src, tgt, 2, 8, 0.05, guess)
    out = [None] * 3

    def work(k):
        for _ in range(3):
            out[k], _ = _run(api, src, tgt, 2, 8, 0.05, guess)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert all(o == alone for o in out)


def test_random_scenes_split_equals_unsplit(api, rs, monkeypatch):
    """Random scenes (clusters, planes, lattices full of exact ties, lines; duplicates, invalid records, queries far
    outside; gates from a fraction of a cell to unbounded): staged kernels, the unscheduled fused kernel and the fused
    kernel with every tile searched by 2 / by 2 and 4 lanes per query give the same bits."""
    from test_nn_fuzz_gpu import scene
    rng = np.random.default_rng(77)
    kinds = ["uniform", "plane", "clusters", "lattice", "line"]
    gates = [0.004, 0.013, 0.05, 0.2, 1e30]
    for it in range(15):
        kind = kinds[it % len(kinds)]
        nt, ns = int(rng.integers(2000, 60000)), int(rng.integers(2000, 40000))
        tgt = scene(rng, kind, nt).astype(np.float32)
        src = (scene(rng, kind, ns) + rng.uniform(-0.02, 0.02, 3)).astype(np.float32)
        if it % 3 == 0:
            tgt[rng.integers(0, nt, nt // 10)] = 0.0
            src[rng.integers(0, ns, ns // 10)] = 0.0
            src[rng.integers(0, ns, ns // 50)] = np.inf
        if it % 4 == 1:
            src[rng.integers(0, ns, ns // 8)] += rng.uniform(-1, 1, 3).astype(np.float32)
        gate = gates[int(rng.integers(0, len(gates)))]
        tc, sc = rs.PointCloud.from_xyz(tgt), rs.PointCloud.from_xyz(src)
        out = []
        for pipeline, env in ((2, {"RSREG_SCHED": "0"}), (0, {"RSREG_SCHED": "0"}),
                              (2, {"RSREG_SCHED_MIN_TILES": "1", "RSREG_SCHED_F2": "0.5", "RSREG_SCHED_F4": "0.5"}),
                              (2, {"RSREG_SCHED_MIN_TILES": "1", "RSREG_SCHED_F2": "1.0", "RSREG_SCHED_F4": "0.0"})):
            for k in ("RSREG_SCHED", "RSREG_SCHED_MIN_TILES", "RSREG_SCHED_F2", "RSREG_SCHED_F4"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            got, _ = _run(api, sc, tc, pipeline, 4, gate)
            out.append(got)
        assert out[0] == out[1] == out[2] == out[3], (it, kind, nt, ns, gate)


def test_carried_schedule_serves_the_first_launch(api, rs, monkeypatch):
    """A schedule built in one alignment serves the next alignments of the same context from their FIRST launch -- the
    unseeded one that reads the source itself --, also for a source with fewer tiles than the schedule knows (its items
    for the missing tiles do nothing) and for one with more (the extra tiles run unsplit behind): same bits as fresh,
    unscheduled contexts."""
    tgt, src = rs.synth.render_frame(0, "N300", "bench"), rs.synth.render_frame(1, "N300", "bench")
    guess = rs.synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
    n = len(src)
    more = np.empty(n + 7001, src.points.dtype)   # more records than the schedule's source had (frame 2's points behind frame 1's)
    more[:n] = src.points
    more[n:] = rs.synth.render_frame(2, "N300", "bench").points[:7001]
    sources = [src,
               rs.PointCloud(src.points[: n - 9000].copy(), width=n - 9000, height=1, is_dense=False),     # fewer tiles
               rs.PointCloud(more, width=n + 7001, height=1, is_dense=False),
               rs.synth.render_frame(2, "N300", "bench")]
    monkeypatch.setenv("RSREG_SCHED", "0")
    base = [_run(api, s, tgt, 2, 4, 0.05, guess)[0] for s in sources]
    monkeypatch.setenv("RSREG_SCHED", "1")
    icp = api.IterativeClosestPoint(api.Context(0))
    icp.params = api.icp_params(max_iterations=4, criteria_mode=1, pipeline_mode=2, max_correspondence_distance=0.05)
    icp.setInputTarget(tgt)
    for k, s in enumerate(sources):
        icp.setInputSource(s)
        out = icp.align(guess)
        r = icp.result
        if icp.grid_info().index_kind != 1:
            pytest.skip("the schedule belongs to the dense-table search")
        got = (bytes(r.transform), bytes(r.sums_last), r.n_correspondences, r.iterations, r.state, r.converged, r.mse,
               np.stack([out.points[c] for c in "xyz"]).tobytes())
        assert got == base[k], k
        if k == 0:
            assert 0 < r.n_scheduled_launches < r.n_nn_launches    # built in this alignment: its first launches run unscheduled
        else:
            assert r.n_scheduled_launches == r.n_nn_launches, (k, r.n_scheduled_launches, r.n_nn_launches)
