#!/bin/bash
# kernel trace of the default bench -> per-step summary, backward view and the per-(kernel, grid) table under gpurun_out/$1/
export TMPDIR=/tmp
out=gpurun_out/${1:-trace_grid}; rm -rf $out/trace; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o bench -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
kt=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/dev/trace_summary.py $kt 6 80 > $out/step_summary.txt
python tools/dev/trace_bwd.py $kt 4 > $out/bwd_summary.txt
python tools/dev/trace_by_grid.py $kt 6 160 > $out/by_grid.txt
head -3 $out/step_summary.txt
rm -rf $out/trace
