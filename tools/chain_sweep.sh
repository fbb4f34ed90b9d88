#!/bin/bash
# the chain workload with K pairs in flight (bench.py --workload chain --in-flight K), one JSON line per K.  usage (gpurun): tools/chain_sweep.sh <out file>
O=$1; : > $O
for K in 0 1 2 3 4 6; do timeout -k 10 300 python bench.py --workload chain --size N300 --frames 16 --in-flight $K --steps 10 --warmup 2 --no-cpu-baseline >> $O 2>> $O.err; done
for K in 0 1 3 4; do timeout -k 10 300 python bench.py --workload chain --size N1M --frames 8 --in-flight $K --steps 6 --warmup 2 --no-cpu-baseline >> $O 2>> $O.err; done
python - $O <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        j = json.loads(l)
        print("chain %d x %d points, %d in flight: %.3f ms per pair, %.3e point-pairs/s, in flight vs sequential max |diff| %s" %
              (j["config"]["n_frames"], j["config"]["points_per_frame"], j["config"]["in_flight"], j["ms_per_pair"], j["value"], j["in_flight_vs_sequential_max_abs_diff"]))
PY
