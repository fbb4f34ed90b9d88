set -o pipefail
mkdir -p gpurun_out/r2l
timeout -k 10 600 python -m pytest tests/test_schemes_gpu.py tests/test_device_clouds_gpu.py tests/test_bench_gpu.py -x -q > gpurun_out/r2l/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2l/pytest.log; tail -5 gpurun_out/r2l/pytest.log
timeout -k 10 400 python bench.py > gpurun_out/r2l/bench.json 2> gpurun_out/r2l/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r2l/bench.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['avg_launch_ms'], d.get('reference_mode'), d.get('host_buffers'))
PY
