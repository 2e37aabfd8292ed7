mkdir -p gpurun_out/r4s
python -m pytest tests -m gpu -x -q > gpurun_out/r4s/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4s/tests.log; tail -3 gpurun_out/r4s/tests.log
bash tools/dev/prof_sq.sh > gpurun_out/r4s/sq.log 2>&1; head -4 gpurun_out/sq/summary.txt | cut -c1-200
python tools/dev/diag_phases.py > gpurun_out/r4s/phases.txt 2>&1; grep -a " ms" gpurun_out/r4s/phases.txt | head -14
