set -o pipefail
cd $GRAFT_REPO_ROOT
for f2 in 0.10 0.25 0.5 1.0; do
echo "f2 $f2: $(RSREG_SCHED_F2=$f2 python tools/iter_times.py N300 | tail -1)" || exit 1
done
for f4 in 0.05 0.2; do
echo "f4 $f4 f2 0.3: $(RSREG_SCHED_F4=$f4 RSREG_SCHED_F2=0.3 python tools/iter_times.py N300 | tail -1)" || exit 1
done
echo "50k f2 0.1 min_tiles 256: $(RSREG_SCHED_MIN_TILES=256 python tools/iter_times.py 50k | tail -1)"
echo "50k default: $(python tools/iter_times.py 50k | tail -1)"
echo "50k f2 1.0 min_tiles 256: $(RSREG_SCHED_MIN_TILES=256 RSREG_SCHED_F2=1.0 python tools/iter_times.py 50k | tail -1)"
