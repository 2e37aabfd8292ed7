#!/bin/bash
# round 6, first GPU call: the de-globalised kernel-form tests + the direct autograd test of the node backward, train.py soaks with the
# per-segment timing, the launcher rehearsal at the largest rank count the pool allows on one card (6)
set -x
mkdir -p gpurun_out/r06
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "rows or longk or bifpn_node_bwd_full or drop_scale" > gpurun_out/r06/t_kernels.log 2>&1; rc=$?
tail -5 gpurun_out/r06/t_kernels.log
[ $rc -eq 0 ] || exit $rc
export MMD_TRAIN_TIMING=1
bash tools/dev/soak.sh 600 '"num_workers": 6, "synthetic_cache": 4' _cache > gpurun_out/r06/soak_cache.txt 2>&1 || { tail -30 gpurun_out/soak_cache/train.log; exit 1; }
grep -a "images/sec\|host seconds\|steady" gpurun_out/soak_cache/train.log
bash tools/dev/soak.sh 600 '"num_workers": 8, "input_pipeline": "raw"' _raw > gpurun_out/r06/soak_raw.txt 2>&1 || { tail -30 gpurun_out/soak_raw/train.log; exit 1; }
grep -a "images/sec\|host seconds\|steady" gpurun_out/soak_raw/train.log
bash tools/dev/soak.sh 600 '"num_workers": 8, "input_pipeline": "raw", "synthetic_cache": 4' _rawcache > gpurun_out/r06/soak_rawcache.txt 2>&1 || { tail -30 gpurun_out/soak_rawcache/train.log; exit 1; }
grep -a "images/sec\|host seconds\|steady" gpurun_out/soak_rawcache/train.log
unset MMD_TRAIN_TIMING
( MMD_FORCE_DEVICE=0 MMD_DIST_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus 6 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r06/launcher_n6.stdout 2> gpurun_out/r06/launcher_n6.stderr; echo "exit status $?" >> gpurun_out/r06/launcher_n6.stdout )
tail -c 1500 gpurun_out/r06/launcher_n6.stdout; tail -5 gpurun_out/r06/launcher_n6.stderr
