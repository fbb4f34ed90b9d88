import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rsreg_amd
from rsreg_amd import api, synth
size = sys.argv[1]
tgt, src = synth.render_frame(0, size, "bench"), synth.render_frame(1, size, "bench")
guess = synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
def run(env):
    for k in list(os.environ):
        if k.startswith("RSREG_SCHED"): del os.environ[k]
    os.environ.update(env)
    icp = api.IterativeClosestPoint(api.Context(0))
    icp.params = api.icp_params(max_iterations=12, criteria_mode=1, pipeline_mode=2, max_correspondence_distance=0.05)
    icp.setInputSource(src); icp.setInputTarget(tgt); icp.align(guess)
    return icp.getFinalTransformation().copy(), bytes(icp.result.sums_last)
base = run({"RSREG_SCHED": "0"})
for f2, f4 in ((0.25, 0.0), (0.5, 0.0), (0.75, 0.0), (1.0, 0.0), (0.0, 0.25), (0.0, 1.0)):
    env = {"RSREG_SCHED_F2": str(f2), "RSREG_SCHED_F4": str(f4), "RSREG_SCHED_MIN_TILES": "1"}
    got = run(env)
    print(size, "f2", f2, "f4", f4, "same:", bool((got[0] == base[0]).all() and got[1] == base[1]), "max|dT|", float(np.abs(got[0]-base[0]).max()))
