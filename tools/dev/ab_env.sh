#!/bin/bash
# alternating bench runs over values of ONE environment variable (same box, same build): ab_env.sh VAR "v1 v2 ..." [rounds] [extra bench args]
var=$1; vals=$2; rounds=${3:-2}; shift 3
for r in $(seq $rounds); do
  for v in $vals; do
    if [ "$v" = "unset" ]; then unset $var; else export $var=$v; fi; python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>&1 >/dev/null | grep -a "timed\|per-step" | sed "s/^/$var=$v  /"
  done
done
