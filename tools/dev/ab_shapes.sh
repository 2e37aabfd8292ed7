#!/bin/bash
# by-shape family times for two settings of one environment variable, same box: ab_shapes.sh VAR v1 v2  -> gpurun_out/shapes_<v>.txt
var=$1; mkdir -p gpurun_out
for v in $2 $3; do
  if [ "$v" = "unset" ]; then unset $var; else export $var=$v; fi
  MMD_PROF_DUMP=gpurun_out/shapes_$v.csv python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/shapes_$v.json 2> gpurun_out/shapes_$v.err || exit 1
  python tools/dev/prof_by_shape.py gpurun_out/shapes_$v.csv 1 400 > gpurun_out/shapes_$v.txt
done
