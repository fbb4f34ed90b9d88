#!/usr/bin/env python3
"""Dev probe: the same launch unscheduled and scheduled, joined by tile (GPU only)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
os.environ["RSREG_DIAG"] = "1"   # (per-launch times, wave stamps and dumps are the diagnostic build's: librsreg_diag.so, csrc/tunables.hpp)
    import rsreg_amd  # noqa: F401
    from rsreg_amd import api, synth
    tgt, src = synth.render_frame(0, "N1M", "bench"), synth.render_frame(1, "N1M", "bench")
    guess = synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
    icp = api.IterativeClosestPoint(api.Context(0, profiling=True))
    icp.params = api.icp_params(max_iterations=int(sys.argv[2]), criteria_mode=1, pipeline_mode=2, max_correspondence_distance=0.05)
    icp.setInputSource(src)
    icp.setInputTarget(tgt)
    icp.align(guess)
    icp.align(guess)
    sys.exit(0)


def run(sched, iters=12):
    path = os.path.join(tempfile.gettempdir(), "rsreg_cmp_%d.bin" % sched)
    env = dict(os.environ, RSREG_WAVE_TIMES=path, RSREG_WAVE_TIMES_LIGHT="1", RSREG_SCHED=str(sched))
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(iters)], env=env, check=True)
    raw = np.fromfile(path, dtype=np.uint64)
    raw = raw[: len(raw) // 16 * 16].reshape(-1, 16)
    raw = raw[(raw[:, 10] >> np.uint64(63)) == 1]
    t0 = raw[:, 0].astype(np.int64)
    us0, us1 = (t0 - t0.min()) / 100.0, (raw[:, 4].astype(np.int64) - t0.min()) / 100.0
    item = raw[:, 12].astype(np.int64)
    return item & 0xffffff, (item >> 28) & 3, us0, us1


tile_a, _, a0, a1 = run(0)
tile_b, lg_b, b0, b1 = run(1)
nt = int(max(tile_a.max(), tile_b.max())) + 1
da = np.zeros(nt)
np.maximum.at(da, tile_a, a1 - a0)
db = np.zeros(nt)
np.maximum.at(db, tile_b, b1 - b0)
sb = np.full(nt, 1e9)
np.minimum.at(sb, tile_b, b0)
lgt = np.zeros(nt, int)
lgt[tile_b] = lg_b
print("span unscheduled %.1f us, scheduled %.1f us" % (a1.max(), b1.max()))
rank = np.argsort(-da)
pos = np.empty(nt, int)
pos[rank] = np.arange(nt)
uns = lgt == 0
worst = np.argsort(-np.where(uns, db, 0))[:25]
print("the unsplit tiles that ran longest in the scheduled launch:")
print("  tile   rank by unscheduled duration (of %d)   unscheduled us   scheduled us   started at" % nt)
for t in worst:
    print("%6d  %6d  %8.1f  %8.1f  %8.1f" % (t, pos[t], da[t], db[t], sb[t]))
for lo, hi in ((0, 0.1), (0.1, 0.2), (0.2, 0.4), (0.4, 0.7), (0.7, 1.0)):
    m = (pos >= lo * nt) & (pos < hi * nt)
    print("rank %.0f-%.0f %%: unscheduled mean %.1f max %.1f | scheduled mean %.1f max %.1f, end mean %.1f max %.1f" %
          (100 * lo, 100 * hi, da[m].mean(), da[m].max(), db[m].mean(), db[m].max(), (sb + db)[m].mean(), (sb + db)[m].max()))
