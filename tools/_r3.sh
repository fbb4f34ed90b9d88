cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3m; mkdir -p $O
for sk in 0 1 3; do for fw in 0 1; do for sc in 1 0; do echo "== DEBUG_SKIP=$sk FAR_WAVE=$fw SCHED=$sc" >> $O/ab.txt; RSREG_DEBUG_SKIP=$sk RSREG_SCHED=$sc RSREG_FAR_WAVE=$fw python tools/iter_times.py N1M 30 2 2>/dev/null | tail -1 >> $O/ab.txt; done; done; done
cat $O/ab.txt
