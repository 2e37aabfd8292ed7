#!/bin/bash
for i in 1 2 3; do
  timeout -k 10 600 python -m pytest tests/test_gpu_step.py -q -s -k "full_size" 2>&1 | grep -a "step 1 (graph\|passed\|failed"
done
