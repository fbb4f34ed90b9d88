#!/bin/bash
# kernel by kernel: the voxel filter of one frame (1 m leaf: what IncrementalICP runs per frame).  usage (gpurun): tools/vox_time.sh <out dir>
O=$GRAFT_REPO_ROOT/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
cat > /tmp/vox_drive.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import rsreg_amd as rs
from rsreg_amd import api
size, leaf = sys.argv[1], float(sys.argv[2])
cloud = rs.synth.render_frame(2, size, "bench")
f = api.ApproximateVoxelGrid(api.default_context())
f.setLeafSize(leaf, leaf, leaf)
f.setInputCloud(cloud)
for _ in range(8):
    out = f.filter()
print(size, leaf, len(cloud), "->", len(out))
PY
for cfg in "N300 1.0" "N1M 1.0" "N300 0.01"; do
  d=$O/kt_$(echo $cfg | tr ' .' '__')
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 /tmp/vox_drive.py $cfg > $d.log 2>&1
  echo "== $cfg: $(grep -- '->' $d.log)"
  python3 - $(find $d -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 8e3
print("   all kernels: %.1f us per filter" % tot)
for r in rows[:16]:
    name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("rsreg::", "")
    print("   %-50s %3d calls %9.1f us each" % (name[:50], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
done
