import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["RSREG_DIAG"] = "1"   # (per-launch times, wave stamps and dumps are the diagnostic build's: librsreg_diag.so, csrc/tunables.hpp)
import rsreg_amd
from rsreg_amd import api, synth
size = sys.argv[1]
iters = int(sys.argv[2])
tgt, src = synth.render_frame(0, size, "bench"), synth.render_frame(1, size, "bench")
guess = synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
def run(env, path):
    for k in list(os.environ):
        if k.startswith("RSREG_SCHED"): del os.environ[k]
    os.environ.update(env)
    os.environ["RSREG_DUMP_SEED"] = path
    icp = api.IterativeClosestPoint(api.Context(0))
    icp.params = api.icp_params(max_iterations=iters, criteria_mode=1, pipeline_mode=2, max_correspondence_distance=0.05)
    icp.setInputSource(src); icp.setInputTarget(tgt); icp.align(guess)
    raw = np.fromfile(path, dtype=np.int32)
    n = int(icp.grid_info().n_source_distinct)
    global PTS
    PTS = raw[5 * n:].view(np.float32).reshape(-1, 4).copy()
    return icp.getFinalTransformation().copy(), raw[:n].copy(), raw[n:5 * n].view(np.float32).reshape(n, 4).copy(), np.frombuffer(bytes(icp.result.sums_last), np.float64).copy()
Ta, sa, qa, sums_a = run({"RSREG_SCHED": "0"}, "/tmp/seed_a.bin")
Tb, sb, qb, sums_b = run({"RSREG_SCHED_F2": "1.0", "RSREG_SCHED_F4": "0", "RSREG_SCHED_MIN_TILES": "1"}, "/tmp/seed_b.bin")
print("iterations", iters, "T same", (Ta == Tb).all(), "sums same", (sums_a == sums_b).all(), "sums diff", np.abs(sums_a - sums_b).max())
d = np.nonzero(sa != sb)[0]
print("queries whose final match differs:", len(d), d[:20], "tiles", np.unique(d // 128)[:20])
dq = np.nonzero((qa != qb).any(axis=1))[0]
print("queries whose final position differs:", len(dq), dq[:20])
P = PTS   # sorted target records: x, y, bits(index), z
px, py, pz, pidx = P[:, 0], P[:, 1], P[:, 3], P[:, 2].view(np.int32)
def d2(q, k):
    dx, dy, dz = np.float32(q[0]) - px[k], np.float32(q[1]) - py[k], np.float32(q[2]) - pz[k]
    return np.float32(np.float32(dx * dx + dy * dy) + dz * dz)
for i in d[:10]:
    q = qa[i]
    dx, dy, dz = q[0] - px, q[1] - py, q[2] - pz
    dd = (dx * dx + dy * dy) + dz * dz
    m = dd.min()
    cand = np.nonzero(dd == m)[0]
    best = cand[np.argmin(pidx[cand])]
    print(i, "tile", i // 128, "lane", i % 128, "match a", sa[i], "d2", d2(q, sa[i]), "idx", pidx[sa[i]], "| b", sb[i], "d2", d2(q, sb[i]), "idx", pidx[sb[i]],
          "| brute force", best, "d2", m, "idx", pidx[best], "ties", len(cand), "x of a/b", px[sa[i]], px[sb[i]])
