set -o pipefail
for v in rows blocks; do for sc in 0 1; do
  echo "far $v sched $sc"
  if [ $v = rows ]; then export RSREG_FAR_ROWS=1; else unset RSREG_FAR_ROWS; fi
  RSREG_SCHED=$sc timeout -k 10 120 python tools/iter_times.py N1M 30 2 2>&1 | tail -1 || exit 1
done; done
unset RSREG_FAR_ROWS
timeout -k 10 300 python tools/wave_timeline.py N1M 30 2>&1 | head -5
