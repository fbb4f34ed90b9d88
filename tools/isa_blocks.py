#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S dump (finds the hot loops).
usage: isa_blocks.py file.s kernel_substring"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % sys.argv[2], l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.set ") or ".Lfunc_end" in lines[i])
blocks, cur = [], {"name": "entry", "valu": 0, "salu": 0, "vmem": 0, "lds": 0, "br": [], "n": 0, "line": start}
for i in range(start + 1, end):
    l = lines[i].strip()
    m = re.match(r"^(\.LBB\w+):", l)
    if m:
        blocks.append(cur)
        cur = {"name": m.group(1), "valu": 0, "salu": 0, "vmem": 0, "lds": 0, "br": [], "n": 0, "line": i}
        continue
    if not l or l.startswith(";") or l.startswith("."):
        continue
    op = l.split()[0]
    cur["n"] += 1
    if op.startswith("v_"):
        cur["valu"] += 1
    elif op.startswith("s_"):
        cur["salu"] += 1
        if "branch" in op:
            cur["br"].append(l.split()[-1])
    elif op.startswith(("buffer_", "global_", "flat_")):
        cur["vmem"] += 1
    elif op.startswith("ds_"):
        cur["lds"] += 1
blocks.append(cur)
idx = {b["name"]: k for k, b in enumerate(blocks)}
for k, b in enumerate(blocks):
    back = [t for t in b["br"] if t in idx and idx[t] <= k]
    print("%-12s line %6d  n=%4d valu=%4d salu=%3d vmem=%2d lds=%2d  %s%s" % (
        b["name"], b["line"], b["n"], b["valu"], b["salu"], b["vmem"], b["lds"], " ".join(b["br"]), "   <== LOOP to " + back[0] if back else ""))
