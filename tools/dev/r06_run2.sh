#!/bin/bash
# round 6, second GPU call: slab-kernel parity, by-shape timing with / without it, step A/B, train soak with tuned teachers, launcher at 4 ranks
set -x
mkdir -p gpurun_out/r06
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "slab or pwconv" > gpurun_out/r06/t_slab.log 2>&1; rc=$?
tail -15 gpurun_out/r06/t_slab.log
[ $rc -eq 0 ] || exit $rc
# by-shape (hipEvents around every launch of one eager single-stream step): with and without the slab kernel
MMD_PROF_DUMP=gpurun_out/r06/shape_slab.csv python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r06/bench_slab.json 2> gpurun_out/r06/bench_slab.err || exit 1
MMD_NO_SLAB=1 MMD_PROF_DUMP=gpurun_out/r06/shape_noslab.csv python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r06/bench_noslab.json 2> gpurun_out/r06/bench_noslab.err || exit 1
python tools/dev/prof_by_shape.py gpurun_out/r06/shape_slab.csv 1 400 > gpurun_out/r06/by_shape_slab.txt
python tools/dev/prof_by_shape.py gpurun_out/r06/shape_noslab.csv 1 400 > gpurun_out/r06/by_shape_noslab.txt
grep -a "f12\|f7 " gpurun_out/r06/by_shape_slab.txt | head -20
echo ---; grep -a "f12\|f7 " gpurun_out/r06/by_shape_noslab.txt | head -20
bash tools/dev/ab_env.sh MMD_NO_SLAB "1 unset" 3 > gpurun_out/r06/ab_slab.txt 2>&1; cat gpurun_out/r06/ab_slab.txt
