export TMPDIR=/tmp
out=gpurun_out/trace_gaps; rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o bench -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
kt=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/dev/trace_gaps.py $kt 2 14 > $out/gaps$1.txt
rm -rf $out/trace
cat $out/gaps$1.txt
