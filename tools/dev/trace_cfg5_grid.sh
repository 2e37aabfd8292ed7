#!/bin/bash
# kernel trace of config 5 (D4 / 768, bf16) -> per-step summary + per-(kernel, grid) table under gpurun_out/$1/
export TMPDIR=/tmp
out=gpurun_out/${1:-trace_cfg5}; rm -rf $out/trace; mkdir -p $out
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o bench -- python3 bench.py --coef 4 --size 768 --precision ${2:-bf16} --no-cpu-baseline --steps 6 --warmup 2 > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
kt=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/dev/trace_summary.py $kt 4 60 > $out/step_summary.txt
python tools/dev/trace_by_grid.py $kt 4 100 > $out/by_grid.txt
head -40 $out/step_summary.txt
rm -rf $out/trace
