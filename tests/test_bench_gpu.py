"""GPU: bench.py prints ONE JSON line with the fields the driver reads, and its multi-rank code
path (process group, native RCCL transport inside rsreg_icp_align, barrier + max-over-ranks
timing) runs -- here with a single rank, which is all a one-GPU box allows."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"]


def run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable] + args, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_json_contract():
    j = run(["bench.py", "--size", "50k", "--steps", "2", "--warmup", "1", "--cpu-iterations", "2"])
    for k in REQUIRED:
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["vs_baseline"] is None
    assert j["value"] > 0 and j["higher_is_better"] is True and j["data"] == "synthetic" and j["dtype"] == "f32"
    assert "workload" in j["config"] and "model" not in j["config"]
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["achieved"] > 0
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    assert j["transform_error_vs_cpu_frobenius"] < 1e-4          # north-star bar


def test_bench_headline_only_runs_nothing_but_the_headline():
    """What rocprofv3 is pointed at (tools/final_profiles.sh): the warm-up and the timed steps, so that a profiler's
    per-kernel means are the headline's -- the contract line with its roofline, none of the other legs."""
    j = run(["bench.py", "--size", "50k", "--steps", "2", "--warmup", "1", "--headline-only"])
    assert j["value"] > 0 and j["roofline"]["avg_launch_ms"] > 0 and j["steps"] == 2
    for leg in ("pipeline1_ms_per_step", "convergence_from_identity", "value_with_events_in_all_steps", "host_buffers", "reference_mode", "cpu_baseline"):
        assert leg not in j, leg


def test_bench_multi_rank_path_with_one_rank():
    j = run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
             str(29600 + os.getpid() % 300), "bench.py", "--gpus", "1", "--size", "50k", "--steps", "2", "--warmup", "1",
             "--no-cpu-baseline", "--force-dist"])
    assert j["n_gpus"] == 1 and j["value"] > 0
    assert j["breakdown_ms_per_step"]["allreduce"] is not None and j["breakdown_ms_per_step"]["allreduce"] >= 0   # ncclAllReduce between events


def test_bench_two_ranks_rehearsed_on_one_gpu():
    """`bench.py --gpus 2` as the driver launches it on an 8-GPU node (BASELINE configs[3]: one pair, source blocks dealt to the
    ranks, the 17 sums all-reduced per iteration, max-over-ranks timing, ONE line from rank 0) -- rehearsed with both ranks on
    device 0 (--devices-per-node 1: gloo, since RCCL refuses two ranks on one device), so that the first multi-GPU box is not
    the first run of the sharding string, the rank-0-only printing and the clean exit of the other rank."""
    port = str(29300 + os.getpid() % 250)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", port, "bench.py", "--gpus", "2", "--size", "50k", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--devices-per-node", "1", "--allow-fallback"],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]        # (torch.distributed.run: non-zero if ANY rank fails)
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["steps"] == 2 and j["value"] > 0
    sh = j["config"]["sharding"]
    assert "dealt round-robin to 2 ranks" in sh and "all-reduce of 17 f64 per iteration" in sh and "gloo" in sh and "rehearsal" in sh
    assert j["config"]["n_src"] == 50000 and j["breakdown_ms_per_step"]["allreduce"] is None     # (the library all-reduced nothing: gloo)
    # the two ranks together matched what one rank matches: the same pair, the same transform to float noise
    one = run(["bench.py", "--size", "50k", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert abs(j["transform_error_vs_ground_truth_frobenius"] - one["transform_error_vs_ground_truth_frobenius"]) < 1e-5
    # without --allow-fallback the same launch refuses to measure another transport than the one the config names: exit code 3
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(int(port) + 1), "bench.py", "--gpus", "2", "--size", "50k", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--devices-per-node", "1"],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode != 0 and not [l for l in r.stdout.split("\n") if l.startswith("{")]


def test_bench_chain_workload():
    """BASELINE configs[4] as bench.py runs it: consecutive pairs, one per GPU at a time, frames resident in HBM."""
    j = run(["bench.py", "--workload", "chain", "--size", "50k", "--steps", "2", "--warmup", "1", "--cpu-iterations", "2"])
    for k in REQUIRED:
        assert k in j, k
    assert j["scaling"] == "weak" and j["n_gpus"] == 1 and j["config"]["n_frames"] == 2 and j["config"]["pairs"] == 1
    assert j["value"] > 0 and j["transform_error_vs_cpu_frobenius"] < 1e-4
    assert j["chain_pose_error_vs_ground_truth_frobenius_max"] < 0.05
    j2 = run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
              str(29900 + os.getpid() % 90), "bench.py", "--gpus", "1", "--workload", "chain", "--size", "50k", "--steps", "2", "--warmup", "1",
              "--no-cpu-baseline", "--force-dist"])
    assert j2["config"]["workload"] == j["config"]["workload"] and j2["value"] > 0


def test_device_resident_inputs_match_host_inputs():
    """rsreg_icp_set_{source,target}_device with record strides 12, 16 and 32 (tests/device_inputs_check.py)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "device_inputs_check.py")], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0 and "device inputs ok" in r.stdout, r.stderr[-3000:]
