"""Run by tests/test_bench_gpu.py in a child process (torch must initialise the GPU before the
engine does when both live in one process: bench.py's import order).  Clouds already in HBM,
record strides 12 / 16 / 32 with xyz first, against the same clouds handed over as host records."""
import os
import sys

import numpy as np
import torch

torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsreg_amd  # noqa: E402
from rsreg_amd import api, synth  # noqa: E402

src, tgt = synth.render_frame(1, "50k", "parity"), synth.render_frame(0, "50k", "parity")
ctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)


def run(setup):
    icp = api.IterativeClosestPoint(ctx)
    icp.params = api.icp_params(max_iterations=5, criteria_mode=1, max_correspondence_distance=0.05)
    setup(icp)
    icp.align()
    r = icp.result
    return bytes(r.transform), bytes(r.sums_last), r.n_correspondences


def host(icp):
    icp.setInputSource(src)
    icp.setInputTarget(tgt)


want = run(host)
for stride in (12, 16, 32):
    def dev(cloud):
        rec = np.zeros((len(cloud), stride // 4), np.float32)
        rec[:, 0], rec[:, 1], rec[:, 2] = cloud.points["x"], cloud.points["y"], cloud.points["z"]
        if stride >= 16:
            rec[:, 3] = 123.0          # whatever follows xyz is ignored
        return torch.from_numpy(rec).cuda()

    ds, dt = dev(src), dev(tgt)
    torch.cuda.synchronize()

    def device(icp):
        icp.setInputSourceDevice(ds.data_ptr(), len(src), stride)
        icp.setInputTargetDevice(dt.data_ptr(), len(tgt), stride)

    got = run(device)
    assert got == want, stride
print("device inputs ok")
