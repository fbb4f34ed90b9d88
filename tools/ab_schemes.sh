#!/bin/bash
# Dev: the C++ scheme runner with two or more builds of librsreg.so on the SAME box, alternating (boxes differ by 10 %).
# usage: tools/ab_schemes.sh <out dir> <name>=<dir with librsreg.so>[@VAR=value] ...     (RSREG_SCHEME_REPS runs per process, five processes each)
O=$1; shift
mkdir -p $O
python tools/cpp_scheme_times.py 50k 2 > /dev/null 2>&1   # builds the runner
for round in 1 2 3 4 5; do
  for spec in "$@"; do
    name=${spec%%=*}; dir=${spec#*=}; var=RSREG_AB_NONE=1
    case "$dir" in *@*) var=${dir#*@}; dir=${dir%%@*};; esac
    env "$var" LD_LIBRARY_PATH=$dir:$LD_LIBRARY_PATH timeout -k 10 300 python tools/cpp_scheme_times.py N300 16 2>&1 | grep "device clouds" | grep -v "run 0" > $O/ab_${name}_$round.txt || exit 1
  done
done
python - $O <<'PY'
import re, glob, sys
import collections
per = collections.defaultdict(lambda: collections.defaultdict(list))   # name -> scheme -> min of every process
for f in sorted(glob.glob(sys.argv[1] + "/ab_*.txt")):
    name = f.split('/')[-1][3:].rsplit('_', 1)[0]
    d = {}
    for l in open(f):
        m = re.search(r"clouds\s+(\w+) run \d+: ([\d.]+) ms", l)
        d.setdefault(m.group(1), []).append(float(m.group(2)))
    for k, v in d.items():
        per[name][k].append(min(v))
for name, d in per.items():   # (a process lands in a fast or a slow mode as a whole: best and median over the processes)
    print("%-12s" % name, "  ".join("%s best %.2f med %.2f [%s]" % (k, min(v), sorted(v)[len(v) // 2], " ".join("%.1f" % x for x in v)) for k, v in d.items()))
PY
