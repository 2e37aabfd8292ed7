#!/bin/bash
# kernel trace of the default bench: per-step summary + backward-only view (GPU box, repo root) -> gpurun_out/trace_step/
export TMPDIR=/tmp
out=gpurun_out/trace_step; rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o bench -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
kt=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/dev/trace_summary.py $kt 6 70 > $out/step_summary.txt
python tools/dev/trace_bwd.py $kt 4 $out/bwd_full.txt $out/step_full.txt > $out/bwd_summary.txt
head -3 $out/step_summary.txt; head -60 $out/bwd_summary.txt
rm -rf $out/trace
