#!/bin/bash
for cfg in "none" "MMD_NO_WG_R32=1" "MMD_NO_SLAB=1"; do
  echo "== $cfg"
  ( [ "$cfg" != none ] && export $cfg; timeout -k 10 500 python -m pytest tests/test_gpu_step.py -x -q -s -k "test_full_size_step_graph_vs_oracle and rgb" 2>&1 | grep -a "gradient cos\|passed\|failed\|Error" )
done
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad" 2>&1 | tail -2
