mkdir -p gpurun_out/r4o
python -m pytest tests -m gpu -x -q > gpurun_out/r4o/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4o/tests.log; tail -3 gpurun_out/r4o/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4o/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r4o/smoke.log
