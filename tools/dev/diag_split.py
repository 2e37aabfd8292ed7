"""Split vs unsplit backward at the two-rank test's size, in one process: which parameter tensors differ?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_step import build
from mm_distillnet_amd.synth import synth_inputs
S, B = int(os.environ.get("S", "256")), 2
batch = {k: v.to("cuda") for k, v in synth_inputs(B, S, seed=40).items()}
res = {}
for name in ("unsplit", "split", "unsplit2"):
    eng, spec = build("pairwise", S)
    if name == "split":
        eng.ar_split = eng._default_split()
    ds = eng.make_drop_scale(B, torch.Generator(device="cuda").manual_seed(3))
    eng.step_body(batch, ds)
    eng.backward_tail()
    torch.cuda.synchronize()
    res[name] = eng.student.ps.export_grads()
for a, b in (("unsplit", "unsplit2"), ("unsplit", "split")):
    worst = []
    for k, v in res[a].items():
        d = (v - res[b][k]).abs().max().item(); m = v.abs().max().item()
        worst.append((d / max(m, 1e-12), d, m, k))
    worst.sort(reverse=True)
    print(a, "vs", b, "largest relative per-tensor differences:")
    for w in worst[:8]:
        print("   %.3e (abs %.3e of %.3e) %s" % w)
