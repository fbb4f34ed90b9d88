set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2prof
mkdir -p $O
cd $R
echo "== bench (default)"; timeout -k 10 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { tail -5 $O/bench_n1.err; exit 1; }
cut -c1-400 $O/bench_n1.json
cd /tmp && export TMPDIR=/tmp
echo "== kernel stats"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/kstats.log 2>&1 || { tail -5 $O/kstats.log; exit 1; }
find $O/kstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_bench.csv
echo "== pmc fetch"
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1 || { tail -5 $O/pmc_fetch.log; exit 1; }
echo "== pmc write"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1 || { tail -5 $O/pmc_write.log; exit 1; }
cd $R
python tools/make_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic.json "round 2" && cat $O/traffic.json | head -40
cd /tmp
echo "== pmc sq"
timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_sq.log 2>&1 || { tail -5 $O/pmc_sq.log; exit 1; }
cd $R
python tools/pmc_summary.py $O/pmc_sq > $O/pmc_sq_counters.txt && cat $O/pmc_sq_counters.txt | cut -c1-300
echo "== timelines"
timeout -k 10 300 python tools/wave_timeline.py N1M 30 > $O/wave_timeline_n1m.txt 2>&1 || { tail $O/wave_timeline_n1m.txt; exit 1; }
RSREG_SCHED=0 timeout -k 10 300 python tools/wave_timeline.py N1M 30 > $O/wave_timeline_n1m_unscheduled.txt 2>&1 || exit 1
timeout -k 10 300 python tools/sched_compare.py > $O/sched_compare_n1m.txt 2>&1 || exit 1
echo "== chain"; timeout -k 10 600 python bench.py --workload chain --steps 5 --warmup 1 > $O/bench_chain_n1.json 2> $O/bench_chain.err || { tail -5 $O/bench_chain.err; exit 1; }
cut -c1-300 $O/bench_chain_n1.json
echo "== configs"; timeout -k 10 900 python tools/bench_configs.py > $O/bench_configs.jsonl 2> $O/bench_configs.err || { tail -5 $O/bench_configs.err; exit 1; }
cut -c1-250 $O/bench_configs.jsonl
