#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean per launch)."""
import collections
import csv
import glob
import re
import sys

path = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
want = sys.argv[2] if len(sys.argv) > 2 else "rsreg"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    k = re.sub(r"<.*>", "", r["Kernel_Name"].split("(")[0]).replace("void ", "").strip()
    if want in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
