cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4a; mkdir -p $O
echo "== base" > $O/iter.txt
timeout -k 10 300 python tools/iter_times.py N1M 30 2 >> $O/iter.txt 2>&1 || { tail -5 $O/iter.txt; exit 1; }
echo "== rows" >> $O/iter.txt
RSREG_ROWS=1 RSREG_ROWS_STATS=1 timeout -k 10 300 python tools/iter_times.py N1M 30 2 >> $O/iter.txt 2>&1 || { tail -20 $O/iter.txt; exit 1; }
echo "== rows nostats" >> $O/iter.txt
RSREG_ROWS=1 timeout -k 10 300 python tools/iter_times.py N1M 30 2 >> $O/iter.txt 2>&1 || { tail -20 $O/iter.txt; exit 1; }
echo "== rows N300" >> $O/iter.txt
RSREG_ROWS=1 timeout -k 10 300 python tools/iter_times.py N300 30 2 >> $O/iter.txt 2>&1 || { tail -20 $O/iter.txt; exit 1; }
echo "== base N300" >> $O/iter.txt
timeout -k 10 300 python tools/iter_times.py N300 30 2 >> $O/iter.txt 2>&1
grep -v "^\[rsreg\] rows" $O/iter.txt | tail -12; grep "rows kernel" $O/iter.txt | tail -4
RSREG_ROWS=1 timeout -k 10 600 python -m pytest tests/test_icp_gpu.py tests/test_nn_fuzz_gpu.py tests/test_nn_full_size_gpu.py tests/test_tile_schedule_gpu.py -q -m gpu -x > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
