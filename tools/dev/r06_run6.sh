#!/bin/bash
# full GPU suite, A/B of the slab kernel, train.py soaks with tuned synthetic teachers (bench.py's workload), launcher rehearsal at 4 ranks
set -x
mkdir -p gpurun_out/r06
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r06/t_all.log 2>&1; rc=$?
tail -5 gpurun_out/r06/t_all.log
[ $rc -eq 0 ] || exit $rc
bash tools/dev/ab_env.sh MMD_NO_SLAB "1 unset" 2 2>&1 | grep timed
export MMD_TRAIN_TIMING=1
bash tools/dev/soak.sh 600 '"num_workers": 6, "synthetic_cache": 4, "synthetic_teacher_candidates": 40' _tuned > gpurun_out/r06/soak_tuned.txt 2>&1 || { tail -30 gpurun_out/soak_tuned/train.log; exit 1; }
grep -a "images/sec\|host seconds\|steady\|Iteration: 600" gpurun_out/soak_tuned/train.log
bash tools/dev/soak.sh 600 '"num_workers": 8, "input_pipeline": "raw", "synthetic_teacher_candidates": 40' _rawtuned > gpurun_out/r06/soak_rawtuned.txt 2>&1 || { tail -30 gpurun_out/soak_rawtuned/train.log; exit 1; }
grep -a "images/sec\|host seconds\|steady\|Iteration: 600" gpurun_out/soak_rawtuned/train.log
unset MMD_TRAIN_TIMING
( MMD_FORCE_DEVICE=0 MMD_DIST_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus 4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r06/launcher_n4.stdout 2> gpurun_out/r06/launcher_n4.stderr; echo "exit status $?" >> gpurun_out/r06/launcher_n4.stdout )
tail -c 1200 gpurun_out/r06/launcher_n4.stdout; tail -3 gpurun_out/r06/launcher_n4.stderr
