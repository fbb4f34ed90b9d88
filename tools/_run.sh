set -o pipefail
echo off; RSREG_SCHED=0 timeout -k 10 120 python tools/iter_times.py N1M 30 2 2>&1 | tail -1 || exit 1
echo xcd; RSREG_SCHED_XCD=1 timeout -k 10 120 python tools/iter_times.py N1M 30 2 2>&1 | tail -1 || exit 1
echo off; RSREG_SCHED=0 timeout -k 10 120 python tools/iter_times.py N1M 30 2 2>&1 | tail -1 || exit 1
echo xcd; RSREG_SCHED_XCD=1 timeout -k 10 120 python tools/iter_times.py N1M 30 2 2>&1 | tail -1 || exit 1
