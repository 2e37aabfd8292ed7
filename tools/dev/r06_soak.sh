#!/bin/bash
# train.py soaks of the final build at the headline shape, bench.py's workload (tuned synthetic teachers): tensor path + cache, raw path
mkdir -p gpurun_out/r06
export MMD_TRAIN_TIMING=1
bash tools/dev/soak.sh 600 '"num_workers": 6, "synthetic_cache": 4, "synthetic_teacher_candidates": 40' _tuned > gpurun_out/r06/soak_tuned.txt 2>&1 || { tail -30 gpurun_out/soak_tuned/train.log; exit 1; }
grep -a "images/sec\|host seconds\|steady\|Iteration: 600" gpurun_out/soak_tuned/train.log
bash tools/dev/soak.sh 600 '"num_workers": 8, "input_pipeline": "raw", "synthetic_teacher_candidates": 40' _rawtuned > gpurun_out/r06/soak_rawtuned.txt 2>&1 || { tail -30 gpurun_out/soak_rawtuned/train.log; exit 1; }
grep -a "images/sec\|host seconds\|steady\|Iteration: 600" gpurun_out/soak_rawtuned/train.log
