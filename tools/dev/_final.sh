#!/bin/bash
# final round-4 evidence besides collect_profiles.sh: by-grid / backward view, phases, SQ counters, the serialised launch timeline, config 5
set -o pipefail
mkdir -p gpurun_out/r4g2
MMD_DIAG_FWD_ONLY=1 python tools/dev/diag_phases.py > gpurun_out/r4g2/phases.txt 2>&1 || { tail -3 gpurun_out/r4g2/phases.txt; exit 1; }
bash tools/dev/prof_sq.sh > /dev/null 2>&1 || exit 1
bash tools/dev/trace_chain.sh r4chain || exit 1
for p in bf16 bf16_hbm fp32; do
  python bench.py --coef 4 --size 768 --precision $p --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r4g2/cfg5_$p.json 2> gpurun_out/r4g2/cfg5_$p.log || { tail -3 gpurun_out/r4g2/cfg5_$p.log; exit 1; }
  grep -a timed gpurun_out/r4g2/cfg5_$p.log
done
bash tools/dev/trace_cfg5_grid.sh r4c5 || exit 1
