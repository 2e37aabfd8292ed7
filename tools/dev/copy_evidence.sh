#!/bin/bash
# copy the summaries of tools/dev/r06_final.sh (gpurun_out/) into profiles/ under the round's names: copy_evidence.sh [prefix, default r06]
p=${1:-r06}; f=gpurun_out/final
cp $f/bench.json profiles/${p}_bench.json
cp $f/kernel_stats.csv profiles/${p}_bench_kernel_stats.csv
cp $f/step_summary.txt profiles/${p}_bench_step_summary.txt
cp $f/bench_under_rocprof.txt profiles/${p}_bench_under_rocprof.txt
cp $f/by_shape.txt profiles/${p}_by_shape.txt
cp $f/pmc_hbm_traffic.csv profiles/${p}_pmc_hbm_traffic.csv
cp $f/pmc_gemm_by_shape.txt profiles/${p}_pmc_gemm_by_shape.txt
cp $f/pmc_node_bwd_variants.txt profiles/${p}_pmc_node_bwd_variants.txt
cp $f/phases.txt profiles/${p}_phases.txt
cp $f/cfg5_bf16.json profiles/${p}_cfg5_bench_bf16.json
cp $f/cfg5_fp32.json profiles/${p}_cfg5_bench_fp32.json
cp $f/d2_bf16.json profiles/${p}_d2_bf16_bench.json
cp gpurun_out/sq/summary.txt profiles/${p}_sq_counters_by_kernel.txt
cp gpurun_out/trace_step/bwd_summary.txt profiles/${p}_backward_view.txt
cp gpurun_out/final_chain/chain.txt profiles/${p}_chain_serialised.txt
