#!/bin/bash
# alternating bench runs over sets of boolean environment flags (same box, same build): ab_flags.sh "none FLAG_A FLAG_A,FLAG_B ..." [rounds]
sets=$1; rounds=${2:-2}
for r in $(seq $rounds); do
  for s in $sets; do
    ( if [ "$s" != none ]; then for f in ${s//,/ }; do export $f=1; done; fi
      python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep -a "timed" | sed "s/^/$s  /" )
  done
done
