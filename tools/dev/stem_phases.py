"""Per-tile timeline of the direct stem weight-gradient kernel (block 0; needs -DMMD_SWSTAMPS)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_distillnet_amd import _lib
call = _lib.call
DEV = "cuda:0"
B, Cin, S, Cout = 8, 8, 512, 32
Kp = 72
OH = S // 2
x = torch.randn(B, Cin, S, S, device=DEV); dz = torch.randn(B * OH * OH, Cout, device=DEV)
dw = torch.zeros(Cout, Kp, device=DEV)
dll = _lib.LIB.load()
ws = torch.empty(int(dll.mmd_stem_wgrad_ws_floats(Cout)), device=DEV)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(3):
    torch.cuda.synchronize(); e0.record()
    call("mmd_stem_conv_bwd_weight", x, dz, dw, ws, B, Cin, S, S, Kp, Cout)
    e1.record(); torch.cuda.synchronize()
st = (ctypes.c_ulonglong * 64)()
assert dll.mmd_sw_stamps(st) == 0
print("event %.1f us" % (e0.elapsed_time(e1) * 1e3))
print("first load issue -> first lstore done: %.2f us" % ((st[1] - st[0]) * 0.01))
for i in range(8):
    b = 4 * i
    print("tile %2d: lstore(+wait) %5.2f  barrier %5.2f  gload issue %5.2f  compute %5.2f  barrier %5.2f" % (
        i, (st[b + 1] - (st[b] if i else st[0])) * 0.01, (st[b + 2] - st[b + 1]) * 0.01, (st[b + 3] - st[b + 2]) * 0.01, (st[b + 4] - st[b + 3]) * 0.01,
        (st[b + 5] - st[b + 4]) * 0.01 if i < 7 else 0.0))
