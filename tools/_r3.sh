cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3g; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "schemes or sharded or bench or configs3 or pcl_pin" > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -3 $O/t.log
python bench.py > $O/bench.json 2>$O/bench.err || { tail $O/bench.err; exit 1; }
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d.get('pipeline1_ms_per_step'), d.get('convergence_from_identity'), d.get('grid_build'), {k:v for k,v in d.get('roofline_issue',{}).items() if k in ('bound_us','frac','avg_launch_us','lane_utilisation')})"
for sz in 250x200 400x313 N300 N1M; do echo "== $sz" >> $O/floors.txt; python tools/iter_times.py $sz 30 2 2>/dev/null | tail -1 >> $O/floors.txt; done
cat $O/floors.txt
