mkdir -p gpurun_out/r4j
python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad_grouped" > gpurun_out/r4j/k.log 2>&1; echo "rc=$?" >> gpurun_out/r4j/k.log; tail -3 gpurun_out/r4j/k.log
python -m pytest tests/test_gpu_net.py tests/test_gpu_step.py -x -q -k "bf16" > gpurun_out/r4j/n.log 2>&1; echo "rc=$?" >> gpurun_out/r4j/n.log; tail -3 gpurun_out/r4j/n.log
for r in 1 2; do
for p in bf16 bf16_hbm; do
python bench.py --coef 4 --size 768 --precision $p --steps 10 --warmup 3 --no-cpu-baseline 2>&1 >/dev/null | grep -a "timed" | sed "s/^/cfg5 $p grouped   /"
MMD_NO_WG_GROUP_BF16=1 python bench.py --coef 4 --size 768 --precision $p --steps 10 --warmup 3 --no-cpu-baseline 2>&1 >/dev/null | grep -a "timed" | sed "s/^/cfg5 $p per-layer /"
done; done
python bench.py --precision bf16 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 >/dev/null | grep -a "timed" | sed "s/^/D2 bf16 grouped /"
MMD_NO_WG_GROUP_BF16=1 python bench.py --precision bf16 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 >/dev/null | grep -a "timed" | sed "s/^/D2 bf16 per-layer /"
