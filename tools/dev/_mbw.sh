mkdir -p gpurun_out/r4h
python -m pytest tests -m gpu -x -q > gpurun_out/r4h/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4h/tests.log; tail -3 gpurun_out/r4h/tests.log
for r in 1 2; do
python bench.py --coef 4 --size 768 --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 >/dev/null | grep -a "timed" | sed "s/^/cfg5 bf16 mbw    /"
MMD_NO_MBW=1 python bench.py --coef 4 --size 768 --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 >/dev/null | grep -a "timed" | sed "s/^/cfg5 bf16 no mbw /"
done
