#!/bin/bash
# dev: the same probe with and without an environment switch, alternating, three rounds
# usage: tools/ab_env_probe.sh VAR=value <python probe and its arguments>
var="$1"; shift
for round in 1 2 3; do
  echo "== default (round $round)"; python "$@" 2>&1 | tail -2
  echo "== $var (round $round)"; env "$var" python "$@" 2>&1 | tail -2
done
