#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -x -q -k "stem" 2>&1 | tail -1
for b in 512 1024 2048; do
  export MMD_STEM_WG_BLOCKS=$b
  out=gpurun_out/stem$b; rm -rf $out; mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o bench -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 > $out/trace.log 2>&1
  kt=$(find $out/trace -name "*kernel_trace.csv" | head -1)
  python tools/dev/trace_by_grid.py $kt 4 400 | grep -a "stem_wgrad" | sed "s/^/blocks=$b  /"
  rm -rf $out/trace
done
unset MMD_STEM_WG_BLOCKS
bash tools/dev/ab_env.sh MMD_STEM_WG_BLOCKS "512 1024 2048" 2 2>&1 | grep -a timed
bash tools/dev/ab_env.sh MMD_NO_STEM_WG_DIRECT "1 unset" 2 2>&1 | grep -a timed
