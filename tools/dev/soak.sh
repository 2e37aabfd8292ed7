#!/bin/bash
# train.py soak at the headline shape (D2, 512², batch 8, synthetic recordings, synthetic teachers): loss trajectory + a validation pass
# usage (GPU box, repo root): bash tools/dev/soak.sh <steps> [extra cfg overrides, e.g. '"input_pipeline": "raw", "num_workers": 8'] [tag]
#   -> gpurun_out/soak<tag>/train.log
steps=${1:-240}
extra=${2:+, $2}
out=$PWD/gpurun_out/soak$3; rm -rf $out; mkdir -p $out
cd $out
ov='{"image_size": 512, "batch_size": 8, "synthetic_length": '$((steps * 8))', "num_epoches": 1, "exp_name": "soak", "resume": "False", "synthetic_cls_bias": -1.0'"$extra"'}'
PYTHONUNBUFFERED=1 timeout -k 10 900 python $GRAFT_REPO_ROOT/train.py --config_file $GRAFT_REPO_ROOT/configs/mm-distillnet.cfg --overwrite "$ov" --max_steps $steps > train.log 2>&1
rc=$?
grep -c "Iteration" train.log; tail -12 train.log
rm -rf $out/soak/*.pth.tar $out/soak/only_parameters* 2>/dev/null
exit $rc
