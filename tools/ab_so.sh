#!/bin/bash
# dev: the same probe with several builds of the library (tools/_build/librsreg_<name>.so), alternating, two rounds
# usage: tools/ab_so.sh "<name> <name> ..." <python probe and its arguments>
names="$1"; shift
for round in 1 2; do
  for n in $names; do
    if [ "$n" = base ]; then so=""; else so="$PWD/tools/_build/librsreg_$n.so"; fi
    echo "== $n (round $round)"
    if [ -z "$so" ]; then python "$@" 2>&1 | tail -2; else RSREG_DIAG=1 RSREG_SO="$so" python "$@" 2>&1 | tail -2; fi   # (RSREG_SO is honoured with RSREG_DIAG=1 only)
  done
done
