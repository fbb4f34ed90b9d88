cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3e; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
tail -3 $O/t.log
python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err || { tail $O/bench.err; exit 1; }
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['breakdown_ms_per_step'], d['reference_mode'])"
