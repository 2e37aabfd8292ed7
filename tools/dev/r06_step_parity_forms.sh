#!/bin/bash
# the full-size whole-step parity test (B = 8, 512 x 512, two optimizer steps against the fp32 CPU oracle) with the printed gradient cosines /
# norm ratios, once per GEMM form: the split form (default) and v_mfma_f32 (MMD_MFMA_F32=1)
for v in split mfma_f32; do
  if [ $v = mfma_f32 ]; then export MMD_MFMA_F32=1; else unset MMD_MFMA_F32; fi
  echo "== $v"
  python -m pytest tests/test_gpu_step.py -m gpu -x -q -s -k "full_size_step_graph_vs_oracle" 2>&1 | grep -a "gradient cos\|passed\|failed\|step [01]"
done
