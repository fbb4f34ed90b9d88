#!/bin/bash
# dev: the reduce's block size (tools/_build/librsreg_rb256.so / _rb1024.so) on the same box, alternating: the headline step and the chain with pairs in flight
for round in 1 2 3; do
  for v in rb256 rb1024; do
    so=$PWD/tools/_build/librsreg_$v.so
    h=$(RSREG_DIAG=1 RSREG_SO=$so timeout -k 10 200 python bench.py --headline-only --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    line="$v round $round: step $h ms;"
    for K in 3 4 6; do
      c=$(RSREG_DIAG=1 RSREG_SO=$so timeout -k 10 200 python bench.py --workload chain --size N300 --frames 16 --in-flight $K --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_pair'])")
      line="$line N300 K=$K $c;"
    done
    c=$(RSREG_DIAG=1 RSREG_SO=$so timeout -k 10 200 python bench.py --workload chain --size N1M --frames 8 --in-flight 4 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_pair'])")
    echo "$line N1M K=4 $c"
  done
done
