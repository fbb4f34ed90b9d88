set -o pipefail
mkdir -p gpurun_out/r2p
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2p/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2p/pytest.log; tail -30 gpurun_out/r2p/pytest.log
