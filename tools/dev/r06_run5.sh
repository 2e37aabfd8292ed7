#!/bin/bash
set -x
mkdir -p gpurun_out/r06
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "slab or mbconv_expand_bwd" 2>&1 | tail -3 || exit 1
bash tools/dev/r06_run3.sh 2>&1 | grep -v "^+" | grep -a "form\|granule\|block 0\|reduction\|combine" 
bash tools/dev/r06_run4.sh 2>&1 | grep -v "^+" | grep -a "f12\|f7 \|timed\|---"
