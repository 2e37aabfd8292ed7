#!/bin/bash
# kernel trace of config 5's architecture (D4 / 768) in one precision mode: per-step summary (GPU box, repo root)
# usage: bash tools/dev/trace_cfg5.sh <precision>     -> gpurun_out/cfg5_<precision>/
export TMPDIR=/tmp
p=${1:-bf16}
out=gpurun_out/cfg5_$p; rm -rf $out; mkdir -p $out
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o bench -- python3 bench.py --coef 4 --size 768 --precision $p --no-cpu-baseline --steps 6 --warmup 2 > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
kt=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/dev/trace_summary.py $kt 4 60 > $out/step_summary.txt
python tools/dev/trace_bwd.py $kt 3 > $out/bwd_summary.txt 2>&1
grep -h '"metric"' $out/trace.log | tail -1 | cut -c1-300
head -45 $out/step_summary.txt
rm -rf $out/trace
