set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2t
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" || exit 1
python bench.py > gpurun_out/r2t/bench_n1.json || exit 1
cut -c1-300 gpurun_out/r2t/bench_n1.json
