#!/bin/bash
for r in 1 2; do
for v in base new; do
  if [ "$v" = "new" ]; then unset MMD_LIB; else export MMD_LIB=$PWD/.ab/base.so; fi
  python bench.py --coef 4 --size 768 --precision fp32 --no-cpu-baseline --steps 10 --warmup 3 2>&1 >/dev/null | grep -a timed | sed "s/^/D4 fp32 $v  /"
done
done
