#!/bin/bash
# Round profiles of the default bench command (run on the GPU box from the repo root):
#   1. bench.py itself (the JSON line with roofline + cpu_baseline)
#   2. rocprofv3 --kernel-trace --stats of the same command  -> per-kernel stats + per-step summary
#   3. two PMC passes (FETCH_SIZE, WRITE_SIZE; counters only with --kernel-trace) -> HBM bytes per kernel family per step
# usage: bash tools/dev/collect_profiles.sh <tag>      (outputs under gpurun_out/<tag>/)
set -o pipefail
tag=${1:-final}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
MMD_PROF_DUMP=$out/prof_dump_clean.csv python bench.py > $out/bench.json 2> $out/bench.log || exit 1
tail -1 $out/bench.json | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
kt=$(find $out/trace -name "*kernel_trace.csv" | head -1); ks=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python tools/dev/trace_summary.py $kt 6 70 > $out/step_summary.txt; head -3 $out/step_summary.txt
cp $ks $out/kernel_stats.csv
grep -h '"metric"' $out/trace.log | tail -1 | cut -c1-200 > $out/bench_under_rocprof.txt
for c in FETCH_SIZE WRITE_SIZE; do
  MMD_PROF_DUMP=$out/prof_dump_$c.csv rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o pmc -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > $out/pmc_$c.log 2>&1 || { tail -5 $out/pmc_$c.log; exit 1; }
done
f=$(find $out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/dev/pmc_summary.py $f $w $out/pmc_node_bwd_variants.txt > $out/pmc_hbm_traffic.csv; cat $out/pmc_hbm_traffic.csv
# the GEMM family's traffic per shape (excess over the algorithmic bytes: where the re-reads are)
python tools/dev/pmc_by_shape.py $f $w $out/prof_dump_FETCH_SIZE.csv > $out/pmc_gemm_by_shape.txt 2>&1; head -30 $out/pmc_gemm_by_shape.txt
# per-shape launch times of the eager roofline step of the plain bench run above (NOT of a counter pass: those launches run ~100 us each)
python tools/dev/prof_by_shape.py $out/prof_dump_clean.csv 1 80 > $out/by_shape.txt 2>&1
# keep only the small summaries (the raw traces are tens of MB)
rm -rf $out/trace $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
ls -la $out
