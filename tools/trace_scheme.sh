#!/bin/bash
# Dev: HIP-API + kernel trace of the C++ scheme runner (last of RSREG_SCHEME_TIME runs), host gaps of the main thread and
# the API / kernel sequence of one frame.  usage: tools/trace_scheme.sh <out dir> <incremental|icp_edge|ndt_edge>
O=$1; mode=$2
mkdir -p $O/frames
trap 'rm -rf $O/frames $O/prof $O/out.*' EXIT
python - $O <<'PY' || exit 1
import os, sys
sys.path.insert(0, os.getcwd())
import rsreg_amd
from rsreg_amd import cloud as cloud_io, synth
for k in range(16):
    cloud_io.save_pcd(sys.argv[1] + "/frames/f%02d.pcd" % k, synth.render_frame(k, "N300", "bench"), binary=True)
PY
python tools/cpp_scheme_times.py 50k 2 > /dev/null 2>&1   # builds the runner
R=$PWD
(cd /tmp && export TMPDIR=/tmp && RSREG_SCHEME_TIME=2 timeout -k 10 300 rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $R/$O/prof -- $R/tests/cpp/_build/scheme_runner $mode $R/$O/out $R/$O/frames/f*.pcd > $R/$O/run.txt 2>&1) || { tail $O/run.txt; exit 1; }
grep " run " $O/run.txt
python - $O <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
api = list(csv.DictReader(open(glob.glob(O + "/prof/*/*hip_api_trace.csv")[0])))
ker = list(csv.DictReader(open(glob.glob(O + "/prof/*/*kernel_trace.csv")[0])))
kname = {r['Correlation_Id']: (r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in ker}
api.sort(key=lambda r: int(r['Start_Timestamp']))
main = collections.Counter(r['Thread_Id'] for r in api).most_common(1)[0][0]
def short(n):
    n = n.replace('void ', '').replace('rocprim::ROCPRIM_400200_NS::detail::', 'rp::').replace('rocprim::ROCPRIM_400200_NS::', 'rp::').replace('(anonymous namespace)::', '').replace('rsreg::', '')
    if 'trampoline_kernel<' in n and 'wrapped_' in n:
        n = 'rp::' + n.split('wrapped_')[1].split('<')[0]
    return n.split('(')[0][:50]
launches = [i for i, r in enumerate(api) if r['Function'] == 'hipLaunchKernel']
per_run = len(launches) // 3
# summary.txt: launches per run (the runner does the scheme three times: two timed repetitions and the one whose result it writes),
# and the kernels of all three by total time
import collections as _c
tot, cnt = _c.Counter(), _c.Counter()
for r in ker:
    nm = short(r['Kernel_Name'])
    tot[nm] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    cnt[nm] += 1
with open(O + "/summary.txt", "w") as f:
    f.write("kernel launches: %d in three runs = %d per run; %.2f ms of kernels per run\n" % (len(ker), len(ker) // 3, sum(tot.values()) / 3e3))
    for nm, t in tot.most_common(28):
        f.write("  %-52s %5d launches per run %9.1f us per run (%.1f us each)\n" % (nm, cnt[nm] // 3, t / 3, t / cnt[nm]))
start, end = launches[-per_run * 9 // 16], launches[-per_run * 6 // 16]   # about three frames of the last run
t0 = int(api[start]['Start_Timestamp'])
quiet = ('hipGetLastError', 'hipGetStreamDeviceId', 'hipGetDevicePropertiesR0600', 'hipDeviceGetAttribute', 'hipSetDevice', 'hipGetDevice')
with open(O + "/sequence.txt", "w") as f:
    prev_end = None
    for r in api[start:end]:
        fn = r['Function']
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if r['Thread_Id'] == main:
            if prev_end is not None and s - prev_end > 30000:
                f.write("          -- main thread outside HIP for %.1f us --\n" % ((s - prev_end) / 1e3))
            prev_end = e
        if fn in quiet:
            continue
        extra = ''
        if fn == 'hipLaunchKernel' and r['Correlation_Id'] in kname:
            k = kname[r['Correlation_Id']]
            extra = ' %-50s gpu %8.1f..%8.1f (%.1f us)' % (short(k[0]), (k[1] - t0) / 1e3, (k[2] - t0) / 1e3, (k[2] - k[1]) / 1e3)
        f.write("%9.1f us tid %s %-22s %7.1f us%s\n" % ((s - t0) / 1e3, r['Thread_Id'][-3:], fn, (e - s) / 1e3, extra))
# gaps.txt: the same window by thread and API call, the GPU's busy share, the caller's thread outside HIP
w0, w1 = int(api[start]['Start_Timestamp']), int(api[end - 1]['End_Timestamp'])
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in ker if int(r['End_Timestamp']) > w0 and int(r['Start_Timestamp']) < w1)
busy, cur_s, cur_e = 0, None, None
for a, b in ks:
    a, b = max(a, w0), min(b, w1)
    if cur_e is None or a > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = a, b
    else:
        cur_e = max(cur_e, b)
if cur_e is not None: busy += cur_e - cur_s
with open(O + "/gaps.txt", "w") as f:
    f.write("window: %.0f us (about three frames of the last run); kernel launches %d; GPU busy (union of the kernels' intervals) %.0f us = %.0f %% of the window\n" %
            ((w1 - w0) / 1e3, len(ks), busy / 1e3, 100.0 * busy / (w1 - w0)))
    by = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in api[start:end]:
        if r['Function'] in quiet: continue
        c = by[r['Thread_Id']][r['Function']]
        c[0] += 1
        c[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    for tid, d in sorted(by.items(), key=lambda kv: -sum(c[0] for c in kv[1].values())):
        f.write("thread ..%s%s: %d API calls\n" % (tid[-3:], " (the caller's thread)" if tid == main else " (a worker of the context)", sum(c[0] for c in d.values())))
        for fn, c in sorted(d.items(), key=lambda kv: -kv[1][1])[:6]:
            f.write("    %-26s %3d calls %8.1f us\n" % (fn, c[0], c[1]))
    out, prev_end = 0.0, None
    for r in api[start:end]:
        if r['Thread_Id'] != main: continue
        s_, e_ = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if prev_end is not None and s_ - prev_end > 30000: out += (s_ - prev_end) / 1e3
        prev_end = e_
    f.write("caller's thread outside HIP for > 30 us at a time: %.0f us\n" % out)
PY
cat $O/gaps.txt
