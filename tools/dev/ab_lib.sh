#!/bin/bash
# alternating bench runs of two builds of the library (same box): ab_lib.sh <base.so> [rounds]   (the other one is the in-tree build)
# MMD_BENCH_ARGS: extra bench.py arguments (another configuration); invalid when the Python side calls a symbol the base library lacks
base=$1; rounds=${2:-3}
for r in $(seq $rounds); do
  for v in base new; do
    if [ $v = base ]; then export MMD_LIB=$PWD/$base; else unset MMD_LIB; fi
    python bench.py --steps ${MMD_AB_STEPS:-30} --warmup 5 --no-cpu-baseline $MMD_BENCH_ARGS 2>&1 >/dev/null | grep -a "timed" | sed "s/^/$v  /"
  done
done
