#!/bin/bash
# the two PMC passes of collect_profiles.sh alone (HBM bytes per kernel family) -> gpurun_out/final/pmc_*.{csv,txt}
set -o pipefail
out=gpurun_out/final; mkdir -p $out; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  MMD_PROF_DUMP=$out/prof_dump_$c.csv rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o pmc -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > $out/pmc_$c.log 2>&1 || { tail -5 $out/pmc_$c.log; exit 1; }
done
f=$(find $out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/dev/pmc_summary.py $f $w $out/pmc_node_bwd_variants.txt > $out/pmc_hbm_traffic.csv; cat $out/pmc_hbm_traffic.csv
python tools/dev/pmc_by_shape.py $f $w $out/prof_dump_FETCH_SIZE.csv > $out/pmc_gemm_by_shape.txt 2>&1
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
