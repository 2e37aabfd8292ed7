#!/bin/bash
# "sizing by removal": alternating bench runs with the named entry points not launched (MMD_DEV_SKIP_CALLS; results wrong, timing only)
# usage: ab_skip.sh "none mmd_se_fc_bwd mmd_affine_act,mmd_chan_pool ..." [rounds]
sets=$1; rounds=${2:-2}
for r in $(seq $rounds); do
  for s in $sets; do
    ( if [ "$s" != none ]; then export MMD_DEV=1 MMD_DEV_SKIP_CALLS=$s; fi
      python bench.py --steps 30 --warmup 5 --no-cpu-baseline --dev-timing 2>&1 >/dev/null | grep -a "timed" | sed "s/^/$s  /" )
  done
done
