#!/bin/bash
# serialised step (every launch alone, in order) under the kernel trace -> gpurun_out/$1/chain.txt
# MMD_BENCH_ARGS: extra bench.py arguments (e.g. "--coef 4 --size 768 --precision bf16" for BASELINE config 5)
export TMPDIR=/tmp MMD_SERIAL=1
out=gpurun_out/${1:-chain}; rm -rf $out/trace; mkdir -p $out
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o bench -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 $MMD_BENCH_ARGS > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
kt=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/dev/trace_chain.py $kt 4 > $out/chain.txt
tail -1 $out/chain.txt
rm -rf $out/trace
