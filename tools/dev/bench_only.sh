#!/bin/bash
# the plain bench record + per-shape launch times of its eager roofline step (first leg of collect_profiles.sh) -> gpurun_out/final/
out=gpurun_out/final; mkdir -p $out
MMD_PROF_DUMP=$out/prof_dump_clean.csv python bench.py > $out/bench.json 2> $out/bench.log || exit 1
python tools/dev/prof_by_shape.py $out/prof_dump_clean.csv 1 80 > $out/by_shape.txt 2>&1
python tools/dev/fam.py rerun < $out/bench.json
