"""The 1 M bench pair at full size: every match of the last fused search launch against scipy's cKDTree (an exact
search that shares nothing with the engine).  `RSREG_DUMP_SEED` makes the diagnostic build's
`rsreg_icp_align` write the position every query matched, the queries themselves and the sorted target records to a file
when it ends."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
scipy_spatial = pytest.importorskip("scipy.spatial")


@pytest.fixture(scope="module")
def api(rs):
    from rsreg_amd import api as a, lib
    lib.build()
    if a.device_count() < 1:
        pytest.fail("no HIP device: the product has no CPU fallback")
    return a


CHILD = r'''
import sys
import numpy as np
sys.path.insert(0, %(root)r)
import rsreg_amd
from rsreg_amd import api, synth
tgt, src = synth.render_frame(0, "N1M", "bench"), synth.render_frame(1, "N1M", "bench")
guess = synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
icp = api.IterativeClosestPoint(api.Context(0))
icp.params = api.icp_params(max_iterations=4, criteria_mode=1, pipeline_mode=2, max_correspondence_distance=%(gate)r)
icp.setInputSource(src)
icp.setInputTarget(tgt)
icp.align(guess)
gi = icp.grid_info()
print("RESULT", gi.index_kind, int(gi.n_source_distinct), int(icp.result.n_scheduled_launches))
'''


@pytest.mark.parametrize("sched", ["default schedule", "every tile split", "unscheduled"])
def test_matches_of_the_1m_pair_against_ckdtree(api, rs, tmp_path, sched):
    """The dump is a diagnostic (csrc/tunables.hpp: the shipped library writes no file): the alignment runs in a child
    process on the diagnostic build of the same sources (RSREG_DIAG=1 -> librsreg_diag.so, the product kernel's own
    instantiation), the matches are checked here."""
    import subprocess
    import sys
    from rsreg_amd import lib
    lib.build_diag()
    gate = 0.05
    path = str(tmp_path / "seed.bin")
    env = dict(os.environ, RSREG_DIAG="1", RSREG_DUMP_SEED=path)
    env.pop("RSREG_SO", None)
    if sched == "unscheduled":
        env["RSREG_SCHED"] = "0"
    elif sched == "every tile split":
        env.update(RSREG_SCHED_MIN_TILES="1", RSREG_SCHED_F4="0.5", RSREG_SCHED_F2="0.5")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": root, "gate": gate}], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    kind, n, n_sched = [int(v) for v in [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()[1:]]
    if kind != 1:
        pytest.skip("the dump belongs to the dense-table search")
    assert (n_sched > 0) == (sched != "unscheduled")
    raw = np.fromfile(path, dtype=np.int32)
    seed = raw[:n]
    q = raw[n:5 * n].view(np.float32).reshape(n, 4)
    rec = raw[5 * n:].view(np.float32).reshape(-1, 4)            # sorted target records: x, y, bits(index), z
    pts = np.stack([rec[:, 0], rec[:, 1], rec[:, 3]], 1)
    pidx = rec[:, 2].view(np.int32)
    dist, nn = scipy_spatial.cKDTree(pts.astype(np.float64)).query(q[:, :3].astype(np.float64), k=1)

    def d2f(a, k):   # FLANN's float32 order
        dx, dy, dz = a[:, 0] - pts[k, 0], a[:, 1] - pts[k, 1], a[:, 2] - pts[k, 2]
        return (dx * dx + dy * dy) + dz * dz

    inside = (q[:, 3] != 0) & (dist <= gate * 0.999)            # (clear of the gate's own rounding)
    assert inside.sum() > 0.9 * n
    assert (seed[inside] >= 0).all(), "a query with a target point inside the gate came back unmatched"
    sel = np.nonzero(inside)[0]
    dk, dt = d2f(q[sel], seed[sel]), d2f(q[sel], nn[sel])
    assert not (dk > dt).any(), "a match is farther than the tree's nearest neighbour"
    tie = dk == dt
    assert (pidx[seed[sel]][tie] <= pidx[nn[sel]][tie]).all(), "among equidistant points the lowest index must win"
