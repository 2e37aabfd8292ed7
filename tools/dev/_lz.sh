mkdir -p gpurun_out/r4k
python -m pytest tests/test_gpu_kernels.py -x -q -k "lazy_operands or bifpn" > gpurun_out/r4k/k.log 2>&1; echo "rc=$?" >> gpurun_out/r4k/k.log; tail -6 gpurun_out/r4k/k.log
python -m pytest tests/test_gpu_net.py tests/test_gpu_step.py -x -q -k "train or golden or full_size or graph" > gpurun_out/r4k/n.log 2>&1; echo "rc=$?" >> gpurun_out/r4k/n.log; tail -4 gpurun_out/r4k/n.log
bash tools/dev/ab_env.sh MMD_NO_LAZY_NODE "unset 1" 3 2>&1 | grep -v per-step
