set -o pipefail
mkdir -p gpurun_out/r2t
timeout -k 10 600 python -m pytest tests/test_icp_gpu.py tests/test_nn_fuzz_gpu.py tests/test_index_paths_gpu.py tests/test_tile_schedule_gpu.py -x -q > gpurun_out/r2t/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/r2t/pytest.log; tail -3 gpurun_out/r2t/pytest.log
[ $rc -eq 0 ] || exit 1
for r in 1 2; do timeout -k 10 120 python tools/iter_times.py N1M 30 2 2>&1 | tail -1 || exit 1; done
timeout -k 10 120 python tools/iter_times.py N300 30 2 2>&1 | tail -1 || exit 1
