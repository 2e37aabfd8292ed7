"""Phase timeline of the depthwise tile kernel on the BatchNorm-1-prologue input-gradient launch (block 0's wall-clock stamps; needs a
-DMMD_DWSTAMPS build: MMD_EXTRA_HIPCC_FLAGS=-DMMD_DWSTAMPS python -m mm_distillnet_amd.build).  usage: dw_phases.py [k] [H] [C]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_distillnet_amd import _lib
call = _lib.call
DEV = "cuda:0"
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
H = int(sys.argv[2]) if len(sys.argv) > 2 else 16
C = int(sys.argv[3]) if len(sys.argv) > 3 else 1248
B, W = 8, H
M = B * H * W
torch.manual_seed(0)
d = lambda t: t.to(DEV).contiguous()
g1, z1, z0 = d(torch.randn(M, C)), d(torch.randn(M, C) * 1.2 + 0.1), d(torch.randn(M, C))
gate, dpool = d(torch.rand(B, C)), d(torch.randn(B, C) * 0.05)
is1, ga1 = torch.rand(C) + 0.5, torch.rand(C) + 0.5
sc1, sh1, mu1 = d(ga1 * is1), d(torch.randn(C) * 0.1), d(torch.randn(C) * 0.2)
is1 = d(is1)
sc0, sh0, mu0, is0 = d(torch.rand(C) + 0.5), d(torch.randn(C) * 0.1), d(torch.randn(C) * 0.2), d(torch.rand(C) + 0.5)
wd = d(torch.randn(k * k, C) / k)
sums1 = torch.zeros(2 * C, dtype=torch.float64, device=DEV)
call("mmd_bn_bwd_reduce", g1, z1, sc1, sh1, mu1, is1, 1, gate, None, dpool, H * W, None, sums1, M, C, None, 0)
dga, dbe = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
dx = torch.empty(M, C, device=DEV)
sums0 = torch.zeros(2 * C, dtype=torch.float64, device=DEV)
dwg = torch.zeros(k * k, C, device=DEV)
junk = torch.empty(64 << 20, device=DEV)
names = ["stage: loads + BatchNorm-1 backward -> LDS", "barrier", "depthwise^T from LDS", "epilogue: bz loads, swish', store, sums", "sums: shuffles, LDS, atomics",
         "weight gradient products", "barrier", "weight gradient: shuffles, LDS, atomics"]
acc = [0.0] * 8
n = 10
ev = 0.0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(n + 2):
    junk.fill_(1.0)
    torch.cuda.synchronize()
    e0.record()
    call("mmd_dwconv_bwd_data_bn1", g1, z1, wd, dx, B, H, W, C, k, sc1, sh1, mu1, is1, sums1, M, gate, dpool, dga, dbe, z0, sc0, sh0, mu0, is0, sums0, None, 0, dwg)
    e1.record()
    torch.cuda.synchronize()
    st = (ctypes.c_ulonglong * 16)()
    assert _lib.LIB.load().mmd_dw_stamps(st) == 0
    if it >= 2:
        ev += e0.elapsed_time(e1) * 1e3
        for i in range(7):
            acc[i] += (st[i + 1] - st[i]) * 0.01
print(f"k {k}  H {H}  C {C}  event {ev / n:.1f} us   block 0: {sum(acc) / n:.1f} us")
for i in range(7):
    print(f"    {names[i]:<48} {acc[i] / n:6.2f} us")
