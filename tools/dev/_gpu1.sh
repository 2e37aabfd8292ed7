mkdir -p gpurun_out/r4b
python -m pytest tests/test_gpu_kernels.py -x -q -k "se_tail or se_path or test_dwconv or mbconv_expand or pool5" -s > gpurun_out/r4b/k.log 2>&1; echo "rc=$?" >> gpurun_out/r4b/k.log
tail -12 gpurun_out/r4b/k.log
grep -c "bit-equal: True" gpurun_out/r4b/k.log; grep -c "bit-equal: False" gpurun_out/r4b/k.log
python -m pytest tests/test_gpu_net.py tests/test_gpu_step.py -x -q > gpurun_out/r4b/net.log 2>&1; echo "rc=$?" >> gpurun_out/r4b/net.log
tail -5 gpurun_out/r4b/net.log
for i in 1 2; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2> gpurun_out/r4b/b_tail_$i.err | cut -c1-250
MMD_NO_SE_TAIL=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2> gpurun_out/r4b/b_notail_$i.err | cut -c1-250
done
grep "timed\|per-step" gpurun_out/r4b/b_*.err
