mkdir -p gpurun_out/r4e
for v in 0 1152 4608 18432; do
  MMD_NODE_FUSE_WIDE_MAXROWS=$v python bench.py --coef 4 --size 768 --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 >/dev/null | grep -a "timed" | sed "s/^/bf16 maxrows=$v /"
done
for v in 0 4608; do
  MMD_NODE_FUSE_WIDE_MAXROWS=$v python bench.py --coef 4 --size 768 --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 >/dev/null | grep -a "timed" | sed "s/^/bf16 maxrows=$v /"
done
python -m pytest tests/test_gpu_net.py -x -q -k "teacher_forced" > gpurun_out/r4e/n.log 2>&1; echo "rc=$?" >> gpurun_out/r4e/n.log; tail -3 gpurun_out/r4e/n.log
