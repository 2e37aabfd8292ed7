mkdir -p gpurun_out/r4p
timeout -k 10 600 python -m pytest tests/test_gpu_net.py tests/test_gpu_step.py -x -q -k "train or golden or full_size or graph or replay or split" > gpurun_out/r4p/n.log 2>&1; echo "rc=$?" >> gpurun_out/r4p/n.log; tail -4 gpurun_out/r4p/n.log
bash tools/dev/ab_env.sh MMD_NO_NODE_BWD_FULL "unset 1" 3 2>&1 | grep -v per-step
