set -o pipefail
mkdir -p gpurun_out/r2s
rm -f gpurun_out/r2s/iter.log
RSREG_CERT_MIN=1 timeout -k 10 600 python -m pytest tests/test_icp_gpu.py tests/test_nn_fuzz_gpu.py tests/test_configs_gpu.py -x -q > gpurun_out/r2s/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2s/pytest.log; tail -5 gpurun_out/r2s/pytest.log
for cfg in "0 2" "1 2" "1 1" "1 4" "1 0.5"; do
  set -- $cfg
  echo "== CERT=$1 SLACK_MM=$2" | tee -a gpurun_out/r2s/iter.log
  RSREG_CERT_STATS=1 RSREG_CERT=$1 RSREG_CERT_SLACK_MM=$2 timeout -k 10 120 python tools/iter_times.py N1M 30 2 2>&1 | tail -3 >> gpurun_out/r2s/iter.log || exit 1
done
cat gpurun_out/r2s/iter.log
