#!/usr/bin/env python3
"""Fold two rocprofv3 PMC passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate runs of the same
bench.py command) into profiles/<name>_traffic.json, per kernel and per launch.

Units and correction follow MI355X_MICROARCH.md (HBM section): the counters are in KB; on gfx950
FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads, WRITE_SIZE is exact.  The
search kernel mixes 16-B, 8-B and 4-B reads, for which the factor is uncalibrated, so both the
raw and the doubled figure are kept; `traffic_bytes_corrected` (2 x FETCH + WRITE) is the upper
bound bench.py quotes.

usage: make_traffic.py <fetch_dir> <write_dir> <out.json> [note]
"""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(d, counter):
    path = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = re.sub(r"<.*>", "", r["Kernel_Name"].split("(")[0]).replace("void ", "").strip()
        if k.startswith("rsreg::"):
            agg[k[len("rsreg::"):]].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {
        "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes of `python3 bench.py --steps 3 "
               "--warmup 1 --headline-only` (N1M pair, 30 iterations; the headline's steps only); per-launch means; KB x 1024 -> bytes; "
               "gfx950 correction per MI355X_MICROARCH.md HBM section: FETCH_SIZE x 2 (exact for 16-B-per-lane "
               "coalesced reads, an upper bound for the narrower reads mixed in), WRITE_SIZE exact; "
               "Infinity-Cache hits are counted, not excluded",
        "note": sys.argv[4] if len(sys.argv) > 4 else "",
        "workload": {"size": "N1M", "pipeline": "fused", "iterations": 30, "max_dist": 0.05},
        "kernels": {},
    }
    for k in sorted(fetch):
        f = sum(fetch[k]) / len(fetch[k])
        w = sum(write[k]) / len(write[k]) if k in write else 0.0
        out["kernels"][k] = {
            "launches": len(fetch[k]),
            "FETCH_SIZE_KB": round(f, 1),
            "WRITE_SIZE_KB": round(w, 1),
            "traffic_bytes_raw": int((f + w) * 1024),
            "traffic_bytes_corrected": int((2 * f + w) * 1024),
        }
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["kernels"].get("k_icp_fused_dense", {})))


if __name__ == "__main__":
    main()
