set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2t
python -m pytest tests/test_ndt_gpu.py -m gpu -x -q > gpurun_out/r2t/t.log 2>&1 || { tail -40 gpurun_out/r2t/t.log; exit 1; }
tail -3 gpurun_out/r2t/t.log
