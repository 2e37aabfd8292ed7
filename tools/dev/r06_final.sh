#!/bin/bash
# final evidence of the round in one call: full GPU suite, smoke, profiles, SQ counters
set -o pipefail
mkdir -p gpurun_out/final
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/final/t_all.log 2>&1; rc=$?; tail -3 gpurun_out/final/t_all.log; [ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
bash tools/dev/final_evidence.sh 2>&1 | tail -8
bash tools/dev/prof_sq.sh > gpurun_out/final/sq.log 2>&1; tail -2 gpurun_out/final/sq.log
