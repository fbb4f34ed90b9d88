#!/bin/bash
# dev: registration() itself in a fresh process (RSREG_SCHEME_COLD=1: after the runtime and the context exist) with two builds of the library, alternating
# usage: tools/ab_cold.sh <name>=<dir with librsreg.so> ...
python tools/cpp_scheme_times.py 50k 2 > /dev/null 2>&1   # builds the runner
D=$(mktemp -d); python - $D <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import rsreg_amd
from rsreg_amd import cloud as cloud_io, synth
for k in range(16):
    cloud_io.save_pcd(sys.argv[1] + "/f%02d.pcd" % k, synth.render_frame(k, "N300", "bench"), binary=True)
PY
for round in 1 2 3 4; do
  for spec in "$@"; do
    name=${spec%%=*}; dir=${spec#*=}
    for mode in incremental icp_edge; do
      LD_LIBRARY_PATH=$dir:$LD_LIBRARY_PATH RSREG_SCHEME_COLD=1 RSREG_SCHEME_TIME=1 RSREG_SCHEME_FRAMES=1 timeout -k 10 120 tests/cpp/_build/scheme_runner $mode $D/out $D/f*.pcd 2>&1 | grep "run 0:\|run 0 frames" | cut -c1-170 | sed "s/^/$name round $round: /"
    done
  done
done
rm -rf $D
