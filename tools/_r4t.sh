cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4t; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_icp_gpu.py tests/test_nn_fuzz_gpu.py tests/test_nn_full_size_gpu.py tests/test_index_paths_gpu.py tests/test_configs_gpu.py tests/test_filters.py -q -m gpu -x --timeout 400 > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for s in N1M N300 50k; do
  timeout -k 10 200 python tools/step_breakdown.py $s 30 source-first 2>&1 | grep -v amdgpu.ids >> $O/steps.txt
  timeout -k 10 200 python tools/ref_mode.py $s 2>&1 | grep -v amdgpu.ids >> $O/ref.txt
done
cat $O/steps.txt $O/ref.txt
