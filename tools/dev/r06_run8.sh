#!/bin/bash
# epilogue loads hoisted in the GEMM kernels: parity + A/B against the previous library
set -x
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "pwconv or slab" 2>&1 | tail -3 || exit 1
MMD_AB_STEPS=30 bash tools/dev/ab_lib.sh .ab/libbase.so 4 2>&1 | grep timed
