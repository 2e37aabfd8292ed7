#!/bin/bash
# alternating bench runs of two builds of the library (same box): ab_lib.sh <base.so> [rounds]   (the other one is the in-tree build)
base=$1; rounds=${2:-3}
for r in $(seq $rounds); do
  for v in base new; do
    if [ $v = base ]; then export MMD_LIB=$PWD/$base; else unset MMD_LIB; fi
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep -a "timed" | sed "s/^/$v  /"
  done
done
