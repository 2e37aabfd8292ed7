set -e
python tools/dev/split3_check.py > gpurun_out/split_f32.txt 2>&1
MMD_SPLIT3=1 python tools/dev/split3_check.py > gpurun_out/split_s3.txt 2>&1
python tools/dev/split3_check.py wide > gpurun_out/split_f32_wide.txt 2>&1
MMD_SPLIT3=1 python tools/dev/split3_check.py wide > gpurun_out/split_s3_wide.txt 2>&1
paste -d'\n' gpurun_out/split_f32.txt gpurun_out/split_s3.txt
echo WIDE
paste -d'\n' gpurun_out/split_f32_wide.txt gpurun_out/split_s3_wide.txt | cut -c1-20,60-
