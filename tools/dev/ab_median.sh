#!/bin/bash
# dev: interleaved A/B of several builds / settings with medians (process-to-process spread of one build is +-2.5 us on
# k_faces: three samples decide nothing).  usage: ab_median.sh <reps> "<label>|<env assignments>|<script + args>" ...
reps=$1; shift
declare -A acc
for ((i = 0; i < reps; i++)); do
  for spec in "$@"; do
    IFS='|' read -r label envs cmd <<< "$spec"
    out=$(env $envs python $cmd 2>/dev/null | tail -1)
    for k in k_fused k_face_count_walk k_faces k_sum; do
      v=$(echo "$out" | grep -o "'$k': [0-9.]*" | grep -o "[0-9.]*$")
      acc["$label $k"]+="$v "
    done
  done
done
for key in "${!acc[@]}"; do
  echo "$key: $(echo ${acc[$key]} | tr ' ' '\n' | sort -n | awk '{a[NR]=$1} END {printf "median %.1f  min %.1f  max %.1f  (n=%d)", a[int((NR+1)/2)], a[1], a[NR], NR}')"
done | sort
