#!/bin/bash
# bf16-native LDS tiles: parity tests of the bf16 modes, config 5 (D4 / 768^2, B = 8) A/B against the previous library, D2 bf16
set -x
mkdir -p gpurun_out/r06
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_net.py -x -q -k "bf16" 2>&1 | tail -4
export MMD_BENCH_ARGS="--coef 4 --size 768 --precision bf16" MMD_AB_STEPS=10
bash tools/dev/ab_lib.sh .ab/libbase.so 2 2>&1 | grep timed
