cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3t; mkdir -p $O
P=$GRAFT_REPO_ROOT/realsense-pointcloud_amd
for rep in 1 2; do for v in base pool; do
  echo "== $v" >> $O/ab.txt
  if [ $v = base ]; then python tools/iter_times.py N1M 30 2 2>&1 | tail -1 >> $O/ab.txt; else RSREG_SO=$P/librsreg_exp_$v.so python tools/iter_times.py N1M 30 2 2>&1 | tail -1 >> $O/ab.txt; fi
done; done
cat $O/ab.txt
cd /tmp; export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
export RSREG_SO=$P/librsreg_exp_pool.so
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/pool -- $B > $O/pool.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $O/pool k_icp_fused_dense | tee -a $O/pmc.txt
