cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3v; mkdir -p $O
python tools/iter_times.py N1M 30 2 2>&1 | tail -2 | cut -c1-260
RSREG_SCHED_CARRY=0 python tools/iter_times.py N1M 30 2 2>&1 | tail -2 | cut -c1-260
python tools/iter_times.py N300 30 2 2>&1 | tail -1
RSREG_SCHED_CARRY=0 python tools/iter_times.py N300 30 2 2>&1 | tail -1
python -m pytest tests -m gpu -x -q -k "tile_schedule or icp_gpu or nn_full or bench or configs" > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -2 $O/t.log
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('pipeline1_ms_per_step'))"; done
