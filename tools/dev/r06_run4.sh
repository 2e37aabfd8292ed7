#!/bin/bash
# by-shape timing with / without the slab kernel + step A/B
set -x
mkdir -p gpurun_out/r06

MMD_PROF_DUMP=gpurun_out/r06/shape_slab.csv python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r06/bench_slab.json 2> gpurun_out/r06/bench_slab.err || exit 1
MMD_NO_SLAB=1 MMD_PROF_DUMP=gpurun_out/r06/shape_noslab.csv python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r06/bench_noslab.json 2> gpurun_out/r06/bench_noslab.err || exit 1
python tools/dev/prof_by_shape.py gpurun_out/r06/shape_slab.csv 1 400 > gpurun_out/r06/by_shape_slab.txt
python tools/dev/prof_by_shape.py gpurun_out/r06/shape_noslab.csv 1 400 > gpurun_out/r06/by_shape_noslab.txt
grep -a "f12\|f7 " gpurun_out/r06/by_shape_slab.txt | head -12
echo ---; grep -a "f12\|f7 " gpurun_out/r06/by_shape_noslab.txt | head -12
bash tools/dev/ab_env.sh MMD_NO_SLAB "1 unset" 3 2>&1 | grep timed
