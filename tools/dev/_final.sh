set -o pipefail
bash tools/dev/collect_profiles.sh r04final > gpurun_out/r04final.log 2>&1 || { tail -5 gpurun_out/r04final.log; exit 1; }
tail -3 gpurun_out/r04final.log
kt=none
bash tools/dev/trace_grid.sh r04final_grid > /dev/null 2>&1
python tools/dev/diag_phases.py > gpurun_out/r04final/phases.txt 2>&1; grep -a " ms" gpurun_out/r04final/phases.txt | head -30
bash tools/dev/prof_teacher.sh > gpurun_out/r04final/teacher.log 2>&1; cp gpurun_out/teacher/summary.txt gpurun_out/r04final/teacher_forward_by_kernel.txt; head -3 gpurun_out/r04final/teacher_forward_by_kernel.txt
