"""How stable are the bf16 D4 teachers' pseudo-labels run to run?  (tests/test_gpu_step.py::test_d4_768_step_vs_oracle[8] compares an eager
step with a graph replay; both use the teachers' own labels)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_step import build, DEV
from mm_distillnet_amd.synth import synth_inputs
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
S, B = 768, 8
eng, spec = build("pairwise", S, precision=prec, coef=4)
batch = {k: v.to(DEV) for k, v in synth_inputs(B, S, seed=33).items()}
ds = eng.make_drop_scale(B, torch.Generator(device=DEV).manual_seed(3))
for i in range(4):
    o = eng.step_body(batch, ds)
    torch.cuda.synchronize()
    print("eager %d: reg %.5f cls %.4f  labels per teacher %s  merged %s" % (i, o["reg"].item(), o["cls"].item(),
          [int(c.sum()) for c in o["cnt_t"]], o["nbox"].cpu().tolist()), flush=True)
eng.capture(batch)
for i in range(3):
    o = eng.replay(batch, ds)
    torch.cuda.synchronize()
    print("replay %d: reg %.5f cls %.4f  labels per teacher %s  merged %s" % (i, o["reg"].item(), o["cls"].item(),
          [int(c.sum()) for c in o["cnt_t"]], o["nbox"].cpu().tolist()), flush=True)
