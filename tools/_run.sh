set -o pipefail
mkdir -p gpurun_out/r2q
timeout -k 10 900 python -m pytest tests/test_nn_fuzz_gpu.py tests/test_index_paths_gpu.py tests/test_sharded_gpu.py -x -q > gpurun_out/r2q/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2q/pytest.log; tail -30 gpurun_out/r2q/pytest.log
