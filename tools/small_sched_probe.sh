#!/bin/bash
# dev: the tile schedule's split fractions on SMALL launches (50 k / 125 k queries: below min_tiles no schedule is built at all)
for size in 50k 400x313; do
  for cfg in "RSREG_AB_NONE=1" "RSREG_SCHED_MIN_TILES=64" "RSREG_SCHED_MIN_TILES=64 RSREG_SCHED_F4=0.25 RSREG_SCHED_F2=0.5" "RSREG_SCHED_MIN_TILES=64 RSREG_SCHED_F4=0.5 RSREG_SCHED_F2=0.5" "RSREG_SCHED_MIN_TILES=64 RSREG_SCHED_F4=1.0 RSREG_SCHED_F2=0.0" "RSREG_SCHED_MIN_TILES=64 RSREG_SCHED_F4=0.0 RSREG_SCHED_F2=1.0"; do
    echo "== $size $cfg: $(env $cfg timeout -k 10 120 python tools/iter_times.py $size 30 2 2>&1 | grep 'avg search' | cut -c1-120)"
  done
done
