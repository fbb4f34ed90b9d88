#!/bin/bash
# A/B of one environment switch, alternating runs so that drift of the box hits both sides alike (dev tool).
# usage: tools/ab_env.sh VAR=VALUE reps -- command ...   (prints the command's last line per run)
SW=$1; REPS=$2; shift 3
for r in $(seq $REPS); do
  echo "A(default) $("$@" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-150)"
  echo "B($SW) $(env $SW "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-150)"
done
