#!/bin/bash
# dev: IncrementalICP's frame clock with and without the streamed result (16 x 307 k frames)
python tools/cpp_scheme_times.py 50k 2 > /dev/null 2>&1
for round in 1 2; do
  for v in 0 1; do
    echo "== RSREG_SCHEME_NO_STREAM=$v (round $round)"
    RSREG_SCHEME_NO_STREAM=$v RSREG_SCHEME_MODES=incremental RSREG_SCHEME_FRAMES=1 RSREG_SCHEME_REPS=4 timeout -k 10 300 python tools/cpp_scheme_times.py N300 16 2>&1 | grep "device clouds" | grep -v "run 0" | cut -c1-220
  done
done
