cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3n; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "index_paths or nn_fuzz or nn_full or icp_gpu or tile_schedule or device_clouds" > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
tail -3 $O/t.log
for rp in 0 1; do
echo "== RSREG_ROCPRIM_SORT=$rp" | tee -a $O/steps.txt
RSREG_ROCPRIM_SORT=$rp python tools/step_breakdown.py N1M 30 source-first | tee -a $O/steps.txt
RSREG_ROCPRIM_SORT=$rp python tools/step_breakdown.py N300 30 source-first | tee -a $O/steps.txt
RSREG_ROCPRIM_SORT=$rp python tools/step_breakdown.py 50k 30 source-first | tee -a $O/steps.txt
RSREG_ROCPRIM_SORT=$rp python tools/ref_mode.py N1M | tee -a $O/steps.txt
RSREG_ROCPRIM_SORT=$rp python tools/ref_mode.py N300 | tee -a $O/steps.txt
RSREG_ROCPRIM_SORT=$rp python tools/ref_mode.py 50k | tee -a $O/steps.txt
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/trace.log 2>&1
