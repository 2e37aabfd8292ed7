mkdir -p gpurun_out/r4r
python -m pytest tests -m gpu -x -q > gpurun_out/r4r/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4r/tests.log; tail -3 gpurun_out/r4r/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4r/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r4r/smoke.log
bash tools/dev/collect_profiles.sh r04final3 > gpurun_out/r04final3.log 2>&1; tail -1 gpurun_out/r04final3.log; head -2 gpurun_out/r04final3/step_summary.txt; cat gpurun_out/r04final3/pmc_hbm_traffic.csv | head -3
bash tools/dev/trace_grid.sh r04final3_grid > /dev/null 2>&1; head -1 gpurun_out/r04final3_grid/bwd_summary.txt
