import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch
from mm_distillnet_amd.model import YetAnotherEfficientDet
from mm_distillnet_amd.synth import synth_inputs
from oracle import effdet_ref as O
from helpers import make_state, grad_state
DEV="cuda"
spec, st = make_state(2, 8, 13, "audio")
m = YetAnotherEfficientDet(compound_coef=2, in_channels=8, device=DEV); m.load_state_dict(st); m.train()
object.__setattr__(m, "_keep", torch.full_like(m._keep, 1.0))
x = synth_inputs(2, 128, seed=25)["audio"]
(c, r, a), f = m(x.to(DEV))
loss = c.sum() * 0.01 + (r ** 2).mean() + sum((u ** 2).mean() for u in f)
loss.backward()
so = grad_state(st)
ones = {b.idx: torch.ones(2) * (1.0 - b.drop_rate) for b in spec.blocks if b.skip}
(co, ro, ao), fo = O.forward(so, x, 2, True, ones)
lo = co.sum() * 0.01 + (ro ** 2).mean() + sum((u ** 2).mean() for u in fo); lo.backward()
print("loss", loss.item(), lo.item(), "cls", (c.cpu()-co).abs().max().item(), "feat", [(u.cpu()-v).abs().max().item() for u,v in zip(f,fo)])
res=[]
for k,p in m.named_parameters():
    ref=so[k].grad; s=ref.abs().max().item()
    res.append(((p.grad.cpu()-ref).abs().max().item()/max(s,1e-30), k, s, p.grad.abs().max().item()))
res.sort(reverse=True)
for t in res[:15]: print(t)
# direct engine grads
eg = m._net.ps.export_grads()
res2=[((eg[k]-so[k].grad).abs().max().item()/max(so[k].grad.abs().max().item(),1e-30),k) for k in eg]
res2.sort(reverse=True); print("engine-level:", res2[:5])
