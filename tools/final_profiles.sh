#!/bin/bash
# The round's committed measurements in one gpurun call (MI355X box): bench line, kernel statistics, PMC passes (HBM traffic,
# SQ counters), step breakdown, reference mode, wave timeline, scheme times.  usage: tools/final_profiles.sh <tag, e.g. r04>
# Everything lands under gpurun_out/<tag>_final/; copy what is to be judged into profiles/.
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_final; mkdir -p $O
export TMPDIR=/tmp
step() { echo "[final_profiles] $1 ($(date +%T))"; }
PART=${2:-all}   # A: the headline's measurements; B: schemes, chain, cold run; all: both (more than one gpurun call's 20 minutes)
if [ "$PART" != B ]; then
step bench
timeout -k 10 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { tail -5 $O/bench_n1.err; exit 1; }
cut -c1-600 $O/bench_n1.json
step "kernel stats"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --headline-only > $O/kt.log 2>&1)
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bench.csv
step "traffic"
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --headline-only > $O/pmc_fetch.log 2>&1)
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --headline-only > $O/pmc_write.log 2>&1)
python tools/make_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic.json "round 6" > $O/traffic.log 2>&1; tail -2 $O/traffic.log
step "SQ counters"
: > $O/pmc_sq_counters.txt
k=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
           "SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"; do
  k=$((k+1))
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_sq$k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --headline-only > $O/pmc_sq$k.log 2>&1)
  python tools/pmc_summary.py $O/pmc_sq$k rsreg >> $O/pmc_sq_counters.txt 2>&1
done
python tools/make_issue.py $O/pmc_sq_counters.txt profiles/r03_valu_issue_microbench.txt $O/issue.json > $O/issue.log 2>&1; tail -2 $O/issue.log
step "launch times, wave timeline"
for s in N1M N300 50k; do echo "== $s" >> $O/launch_times.txt; timeout -k 10 200 python tools/iter_times.py $s 30 2 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $O/launch_times.txt; done
for s in 125k; do echo "== 125 k points (400 x 313)" >> $O/launch_times.txt; timeout -k 10 200 python tools/iter_times.py 400x313 30 2 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $O/launch_times.txt; done
RSREG_WAVE_TIMELINE_JSON=$O/wave_timeline.json timeout -k 10 300 python tools/wave_timeline.py N1M 30 > $O/wave_timeline_n1m.txt 2>&1
step "index build and source load, kernel by kernel"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d $O/kb -o b -- python3 $GRAFT_REPO_ROOT/tools/build_kernels.py N1M 20 > $O/kb.log 2>&1)
python tools/kstats.py $O/kb > $O/build_kernels.txt 2>&1 < /dev/null; head -24 $O/build_kernels.txt
step "step breakdown, reference mode"
for s in N1M N300 50k; do
  timeout -k 10 200 python tools/step_breakdown.py $s 30 source-first 2>&1 | grep -v amdgpu.ids >> $O/step_breakdown.txt
  timeout -k 10 200 python tools/ref_mode.py $s 2>&1 | grep -v amdgpu.ids >> $O/reference_mode.txt
done
echo "== python tools/cpp_pair_time.py: the same pair through tools/cpp/pair_time.cpp, a C++ caller of the C ABI (no Python between the calls)" >> $O/reference_mode.txt
timeout -k 10 300 python tools/cpp_pair_time.py 2>&1 | grep -v amdgpu.ids >> $O/reference_mode.txt
echo "== python tools/cpp_pair_time.py --host: the literal call surface, host clouds in, 4x4 + aligned cloud out (rsreg_ctx_host_timing's breakdown inside)" >> $O/reference_mode.txt
timeout -k 10 300 python tools/cpp_pair_time.py N1M N300 50k --host 2>&1 | grep -v amdgpu.ids >> $O/reference_mode.txt
timeout -k 10 200 python tools/small_align.py 300 2>&1 | grep -v amdgpu.ids >> $O/reference_mode.txt
timeout -k 10 200 python tools/small_ndt.py 100 2>&1 | grep -v amdgpu.ids >> $O/reference_mode.txt
cat $O/step_breakdown.txt $O/reference_mode.txt
fi
if [ "$PART" != A ]; then
step "schemes"
RSREG_SCHEME_REPS=3 timeout -k 10 900 bash tools/ab_schemes.sh $O/ab cur=realsense-pointcloud_amd > $O/cpp_scheme_times.txt 2>&1; cat $O/cpp_scheme_times.txt
echo "== RSREG_SCHEME_FRAMES=1 RSREG_SCHEME_REPS=4 python tools/cpp_scheme_times.py N300 16: ms until the loop starts | every frame's pass through the loop | until the merged cloud is complete on the host" >> $O/cpp_scheme_times.txt
RSREG_SCHEME_FRAMES=1 RSREG_SCHEME_REPS=4 timeout -k 10 300 python tools/cpp_scheme_times.py N300 16 2>&1 | grep "device clouds" | grep -v "finish:" | cut -c1-330 >> $O/cpp_scheme_times.txt
for m in incremental icp_edge ndt_edge; do
  timeout -k 10 300 bash tools/trace_scheme.sh gpurun_out/${TAG}_final/trace_$m $m > $O/trace_$m.log 2>&1 < /dev/null   # (a path relative to the repository)
  echo "== $m (tools/trace_scheme.sh: rocprofv3 --hip-runtime-trace --kernel-trace of tests/cpp/scheme_runner.cpp, 16 x 307 k frames)" >> $O/scheme_kernel_stats.txt
  cat $O/trace_$m/summary.txt >> $O/scheme_kernel_stats.txt
done
head -12 $O/scheme_kernel_stats.txt
timeout -k 10 300 python tools/scheme_times.py N300 16 2>&1 | grep -v amdgpu.ids > $O/scheme_times.txt; tail -8 $O/scheme_times.txt
step "other workloads"
timeout -k 10 300 python bench.py --workload chain --steps 5 --warmup 1 > $O/bench_chain_n1.json 2> $O/bench_chain.err; cut -c1-300 $O/bench_chain_n1.json
step "chain of 16 frames, K pairs in flight (BASELINE configs[4] on one GPU)"
: > $O/bench_chain_in_flight.jsonl
for K in 0 1 2 3 4 6; do timeout -k 10 300 python bench.py --workload chain --size N300 --frames 16 --in-flight $K --steps 10 --warmup 2 --no-cpu-baseline >> $O/bench_chain_in_flight.jsonl 2>> $O/bench_chain.err; done
for K in 0 1 3 4; do timeout -k 10 300 python bench.py --workload chain --size N1M --frames 8 --in-flight $K --steps 6 --warmup 2 --no-cpu-baseline >> $O/bench_chain_in_flight.jsonl 2>> $O/bench_chain.err; done
timeout -k 10 300 python bench.py --workload chain --size N300 --frames 16 --in-flight 4 --steps 10 --warmup 2 --cpu-iterations 10 > $O/bench_chain16_k4.json 2>> $O/bench_chain.err
python - $O/bench_chain_in_flight.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        j = json.loads(l)
        print("chain %d x %d points, %d in flight: %.3f ms per pair, %.3e point-pairs/s, in flight vs sequential max |diff| %s" %
              (j["config"]["n_frames"], j["config"]["points_per_frame"], j["config"]["in_flight"], j["ms_per_pair"], j["value"], j["in_flight_vs_sequential_max_abs_diff"]))
PY
step "C++ ChainRegistrar from host frames (tests/cpp/scheme_runner.cpp chain)"
python - $O <<'PY'
import os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd())
import rsreg_amd
from rsreg_amd import cloud as cloud_io, synth
exe = "tests/cpp/_build/scheme_runner"
with tempfile.TemporaryDirectory() as d:
    paths = []
    for k in range(16):
        p = os.path.join(d, "f%02d.pcd" % k)
        cloud_io.save_pcd(p, synth.render_frame(k, "N300", "bench"), binary=True)
        paths.append(p)
    with open(sys.argv[1] + "/cpp_chain_times.txt", "w") as f:
        for iters in ("0", "30"):
            for K in ("1", "2", "3", "4"):
                r = subprocess.run([exe, "chain", os.path.join(d, "out")] + paths, env=dict(os.environ, RSREG_SCHEME_TIME="4", RSREG_CHAIN_IN_FLIGHT=K, RSREG_CHAIN_ITERATIONS=iters),
                                   stderr=subprocess.PIPE, text=True)
                for line in r.stderr.splitlines():
                    if "ms per pair" in line and "run 0" not in line:
                        f.write("%s iterations, %s\n" % ("reference parameters (1)" if iters == "0" else iters + " fixed", line))
print(open(sys.argv[1] + "/cpp_chain_times.txt").read())
PY
step "cold run, ISA line"
timeout -k 10 600 python tools/cold_run.py N300 16 3 > $O/cold_run.txt 2>&1; grep "run 0:" $O/cold_run.txt | cut -c1-160
bash tools/isa.sh > $O/isa.txt 2>&1; cat $O/isa.txt
timeout -k 10 600 python tools/bench_configs.py > $O/bench_configs.jsonl 2> $O/bench_configs.err; cut -c1-250 $O/bench_configs.jsonl
fi
step "clean up"
rm -rf $O/kt $O/kb $O/pmc_fetch $O/pmc_write $O/pmc_sq? $O/ab $O/trace_*/sequence.txt
ls -la $O
