set -o pipefail
mkdir -p gpurun_out/r2t
for c in 0 1 2 4 0; do
echo "sort cfg $c"
RSREG_SORT_CFG=$c timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2t/b.json 2> gpurun_out/r2t/b.err || { tail -20 gpurun_out/r2t/b.err; exit 1; }
python -c "
import json
d=json.load(open('gpurun_out/r2t/b.json')); print(d['ms_per_step'], d['breakdown_ms_per_step'])"
done
