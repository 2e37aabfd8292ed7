"""Stand-alone time of mmd_maxpool_same_bwd_acc2 (the P5 -> P6 -> P7 pools' backward).  usage: one_maxpool_bwd.py [PH] [C] [B]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_distillnet_amd import _lib
call = _lib.call
PH = int(sys.argv[1]) if len(sys.argv) > 1 else 16
C = int(sys.argv[2]) if len(sys.argv) > 2 else 112
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
DEV = "cuda:0"
OH = (PH + 1) // 2
src = torch.randn(B * PH * PH, C, device=DEV); dout = torch.randn(B * OH * OH, C, device=DEV); dst = torch.zeros(B * PH * PH, C, device=DEV)
z = torch.randn(B * PH * PH, C, device=DEV); mu = torch.randn(C, device=DEV); istd = torch.rand(C, device=DEV) + 0.5
sums = torch.zeros(2 * C, dtype=torch.float64, device=DEV)
junk = torch.empty(64 << 20, device=DEV)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = 0.0
for it in range(12):
    junk.fill_(1.0); torch.cuda.synchronize()
    e0.record()
    call("mmd_maxpool_same_bwd_acc2", src, dout, dst, None, 0, 0, 1, B, PH, PH, C, z, mu, istd, sums)
    e1.record(); torch.cuda.synchronize()
    if it >= 2: tot += e0.elapsed_time(e1) * 1e3
print(f"maxpool bwd  B {B}  source {PH}^2  C {C}: {tot / 10:.1f} us per launch (event time, ~12 us of it launch overhead)")
