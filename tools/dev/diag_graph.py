import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch
from mm_distillnet_amd.synth import synth_inputs
from test_gpu_step import build
DEV = "cuda"
S, B = 128, 2
batch = {k: v.to(DEV) for k, v in synth_inputs(B, S, seed=5).items()}
ea, spec = build("pairwise", S); eb, _ = build("pairwise", S); ec, _ = build("pairwise", S)
g = torch.Generator(device=DEV).manual_seed(1)
scales = [ea.make_drop_scale(B, g) for _ in range(3)]
ec.capture(batch)
def rel(a, b): return ((a - b).abs().max() / b.abs().max()).item()
for i, ds in enumerate(scales):
    ea.step_body(batch, ds); eb.step_body(batch, ds)
    ec.set_drop_scale(ds); ec.g_main.replay()
    torch.cuda.synchronize()
    ga, gb, gc = ea.student.ps.grad, eb.student.ps.grad, ec.student.ps.grad
    print(i, "grad eager-eager", rel(ga, gb), "eager-graph", rel(ga, gc), "gnorm", ga.norm().item(), gc.norm().item(),
          "losses", ea.out["cls"].item(), ec.out["cls"].item(), ea.out["reg"].item(), ec.out["reg"].item(),
          "nbox", ea.out["nbox"].tolist(), ec.out["nbox"].tolist())
    ea.optimizer_body(); eb.optimizer_body(); ec.g_opt.replay()
    torch.cuda.synchronize()
    pa, pb, pc = ea.student.ps.flat, eb.student.ps.flat, ec.student.ps.flat
    print(i, "param eager-eager max", (pa - pb).abs().max().item(), "frac>2e-5", ((pa - pb).abs() > 2e-5).float().mean().item(),
          "eager-graph max", (pa - pc).abs().max().item(), "frac", ((pa - pc).abs() > 2e-5).float().mean().item(),
          "steps", ea.adam_main[0].item(), ec.adam_main[0].item(), ea.adam_head[0].item(), ec.adam_head[0].item())
