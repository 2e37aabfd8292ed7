"""Host cost of replaying the step graph: time until hipGraphLaunch returns vs time until the GPU is done.
   usage: python tools/dev/diag_launch.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_distillnet_amd.arch import make_spec
from mm_distillnet_amd.synth import synth_state, synth_inputs
from mm_distillnet_amd.step import DistillEngine, StepConfig

dev = "cuda:0"; S, B = 512, 8
torch.cuda.set_device(0)
specs = {"rgb": make_spec(2, 3), "depth": make_spec(2, 3), "thermal": make_spec(2, 1)}
sspec = make_spec(2, 8)
eng = DistillEngine(sspec, specs, dev, StepConfig(image_size=S))
if os.environ.get("SPLIT"):
    eng.ar_split = eng._default_split()
eng.load(synth_state(sspec, seed=9), {k: synth_state(s, seed=i + 1) for i, (k, s) in enumerate(specs.items())})
batch = {k: v.to(dev) for k, v in synth_inputs(B, S, seed=24).items()}
eng.capture(batch)
for _ in range(3):
    eng.replay()
torch.cuda.synchronize()
hl, tt = [], []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.g_main.replay()
    t1 = time.perf_counter()
    if eng.g_tail is not None:
        eng.g_tail.replay()
    eng.g_opt.replay()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    hl.append((t1 - t0) * 1e3); tt.append((t3 - t0) * 1e3)
hl.sort(); tt.sort()
print("host g_main.replay() returns after: median %.3f ms (min %.3f)   synced step: median %.3f ms (min %.3f)" % (
    hl[len(hl) // 2], hl[0], tt[len(tt) // 2], tt[0]), flush=True)
t0 = time.perf_counter()
for _ in range(30):
    eng.replay()
torch.cuda.synchronize()
print("back-to-back: %.3f ms/step" % ((time.perf_counter() - t0) / 30 * 1e3), flush=True)
