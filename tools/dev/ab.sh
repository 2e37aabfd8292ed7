#!/bin/bash
# usage: ab.sh "<label>=<env assignments>" ...   ; alternating runs, 2 rounds
run() { env $2 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['per_step']['median_ms'], d['per_step']['min_ms'])"; }
for r in 1 2; do
  for spec in "$@"; do
    label="${spec%%=*}"; envs="${spec#*=}"
    run "$label" "$envs"
  done
done
