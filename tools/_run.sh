set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2final
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
[ $rc -eq 0 ] || { tail -60 $O/pytest_gpu.log; exit 1; }
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
timeout -k 10 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { tail -5 $O/bench_n1.err; exit 1; }
cut -c1-900 $O/bench_n1.json
