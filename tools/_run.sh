set -o pipefail
mkdir -p gpurun_out/r2j
rm -f gpurun_out/r2j/iter.log
timeout -k 10 600 python -m pytest tests/test_icp_gpu.py tests/test_nn_fuzz_gpu.py tests/test_index_paths_gpu.py -x -q > gpurun_out/r2j/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2j/pytest.log; tail -3 gpurun_out/r2j/pytest.log
for c in 0 2 4 8 16 64; do
  echo "== FAR_COOP=$c" | tee -a gpurun_out/r2j/iter.log
  RSREG_FAR_COOP=$c timeout -k 10 120 python tools/iter_times.py N1M 30 2 2>&1 | tail -1 >> gpurun_out/r2j/iter.log || exit 1
done
cat gpurun_out/r2j/iter.log
