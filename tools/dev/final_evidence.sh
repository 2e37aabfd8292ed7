#!/bin/bash
# round-end evidence in one GPU call: profiles of the default bench, the serialised chain, the phase split, the other configurations' bench lines
set -o pipefail
out=gpurun_out/final; mkdir -p $out
bash tools/dev/collect_profiles.sh final > $out/collect.log 2>&1 || { tail -5 $out/collect.log; exit 1; }
bash tools/dev/trace_chain.sh final_chain > $out/chain.log 2>&1 || { tail -5 $out/chain.log; exit 1; }
bash tools/dev/trace_step.sh > $out/trace_step.log 2>&1 || { tail -5 $out/trace_step.log; exit 1; }
MMD_DIAG_FWD_ONLY=1 timeout -k 10 300 python tools/dev/diag_phases.py > $out/phases.txt 2> $out/phases.err || { tail -5 $out/phases.err; exit 1; }
for p in bf16 fp32; do
  python bench.py --coef 4 --size 768 --precision $p --no-cpu-baseline --steps 20 --warmup 3 2> $out/cfg5_$p.err | tail -1 > $out/cfg5_$p.json || exit 1
  grep -a timed $out/cfg5_$p.err
done
python bench.py --precision bf16 --no-cpu-baseline --steps 30 --warmup 5 2> $out/d2_bf16.err | tail -1 > $out/d2_bf16.json; grep -a timed $out/d2_bf16.err
echo done
