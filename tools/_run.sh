set -o pipefail
mkdir -p gpurun_out/r2p
timeout -k 10 500 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2p/bench_quick.json 2> gpurun_out/r2p/bench_quick.err || { tail -5 gpurun_out/r2p/bench_quick.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2p/bench_quick.json"))
print(d["value"], d["ms_per_step"], d["breakdown_ms_per_step"], d["roofline"]["avg_launch_ms"])
PY
timeout -k 10 600 python -m pytest tests/test_icp_gpu.py tests/test_index_paths_gpu.py tests/test_nn_fuzz_gpu.py -x -q 2>&1 | tail -3
