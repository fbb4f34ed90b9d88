#!/bin/bash
# dev: the C++ scheme runner under different numbers of hardware queues (GPU_MAX_HW_QUEUES, read by the HIP runtime when it starts; default 4)
python tools/cpp_scheme_times.py 50k 2 > /dev/null 2>&1   # builds the runner
for round in 1 2 3; do
  for q in 4 8 16 2; do
    echo "== GPU_MAX_HW_QUEUES=$q (round $round)"
    GPU_MAX_HW_QUEUES=$q RSREG_SCHEME_REPS=5 timeout -k 10 300 python tools/cpp_scheme_times.py N300 16 2>&1 | grep "device clouds" | grep " run [1-9]:" | cut -c1-60
  done
done
