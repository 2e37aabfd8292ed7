python -m pytest tests/test_gpu_kernels.py -x -q -k "lazy_operands" 2>&1 | tail -1
for r in 1 2 3; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep -a "timed" | sed "s/^/round 4 build          /"
MMD_NO_LAZY_NODE=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep -a "timed" | sed "s/^/no lazy nodes          /"
done
MMD_DIAG_FWD_ONLY=1 python tools/dev/diag_phases.py 2>&1 | grep -a "student forward"
MMD_NO_LAZY_NODE=1 MMD_DIAG_FWD_ONLY=1 python tools/dev/diag_phases.py 2>&1 | grep -a "student forward"
