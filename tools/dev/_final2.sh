mkdir -p gpurun_out/r04cfg5
for p in bf16 bf16_hbm fp32; do
  python bench.py --coef 4 --size 768 --precision $p --steps 10 --warmup 3 --no-cpu-baseline 2> gpurun_out/r04cfg5/bench_$p.log > gpurun_out/r04cfg5/bench_$p.json; grep -a "timed" gpurun_out/r04cfg5/bench_$p.log | sed "s/^/cfg5 $p /"
done
bash tools/dev/trace_cfg5_grid.sh r04cfg5_trace bf16 > /dev/null 2>&1; head -3 gpurun_out/r04cfg5_trace/step_summary.txt
bash tools/dev/prof_sq.sh > gpurun_out/r04cfg5/sq.log 2>&1; head -5 gpurun_out/sq/summary.txt
python bench.py --precision bf16 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 >/dev/null | grep -a "timed" | sed "s/^/D2 bf16 /"
