set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r2t/bp; mkdir -p $R/gpurun_out/r2t
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2t/bp -- python3 $R/bench.py --steps 8 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r2t/bp.log 2>&1 || { tail $R/gpurun_out/r2t/bp.log; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/r2t/bp/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print('   ', r['Name'][:40], 'calls', r['Calls'], 'avg us', float(r['AverageNs'])/1e3, 'min', float(r['MinNs'])/1e3)
PY
