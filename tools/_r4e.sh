cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4e; mkdir -p $O
for s in N1M N300 50k; do
  timeout -k 10 200 python tools/step_breakdown.py $s 30 source-first >> $O/steps.txt 2>&1 || { tail -20 $O/steps.txt; exit 1; }
  RSREG_ROCPRIM_SORT=1 timeout -k 10 200 python tools/step_breakdown.py $s 30 source-first 2>&1 | sed 's/^/rocprim driver: /' >> $O/steps.txt
  timeout -k 10 200 python tools/ref_mode.py $s >> $O/ref.txt 2>&1
  RSREG_ROCPRIM_SORT=1 timeout -k 10 200 python tools/ref_mode.py $s 2>&1 | sed 's/^/rocprim driver: /' >> $O/ref.txt
done
cat $O/steps.txt $O/ref.txt
timeout -k 10 900 python -m pytest tests/test_icp_gpu.py tests/test_nn_fuzz_gpu.py tests/test_device_clouds_gpu.py tests/test_configs_gpu.py tests/test_index_paths_gpu.py -q -m gpu -x > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
