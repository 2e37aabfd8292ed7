"""Phase timeline of the whole-node BiFPN FORWARD kernel (eval and train forms) on a small map (block 0's wall-clock stamps; needs a
-DMMD_NODE_TIMING build).  usage: node_fwd_phases.py [H] [B]"""
import ctypes, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_distillnet_amd import _lib
call = _lib.call
DEV = "cuda:0"
H = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
C, W = 112, H
g = lambda t: t.to(DEV).contiguous()
names = ["weights / taps issue + operand loads + windows + fuse + swish -> LDS", "barrier", "depthwise 3x3 (LDS)", "barrier", "park weights in LDS",
         "MFMA + epilogue stores", "train: sums"]
for train in (False, True):
    for mode in ("td", "bu", "p7"):
        M = B * H * W
        has1, hasu, hasp = mode == "bu", mode == "td", mode in ("bu", "p7")
        torch.manual_seed(0)
        in0 = g(torch.randn(M, C)); in1 = g(torch.randn(M, C)) if has1 else None
        up = g(torch.randn(M // 4, C)) if hasu else None
        pl = g(torch.randn(4 * M, C) - 1.0) if hasp else None
        theta = g(torch.tensor([0.7, 1.3, 0.4][:2 if mode != "bu" else 3]))
        wd = g(torch.randn(9, C) / 3); wp = g(torch.randn(C, C) / math.sqrt(C)); bias = g(torch.randn(C) * 0.1)
        sc, sh = g(torch.rand(C) + 0.5), g(torch.randn(C) * 0.1)
        y = torch.empty(M, C, device=DEV); zd = torch.empty(M, C, device=DEV); st = torch.zeros(2 * C, dtype=torch.float64, device=DEV)
        junk = torch.empty(64 << 20, device=DEV)
        acc = [0.0] * 7
        n = 10
        ev = 0.0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(n + 2):
            junk.fill_(1.0); torch.cuda.synchronize(); e0.record()
            if train:
                call("mmd_bifpn_node_fwd_fused_train", in0, in1, up, pl, theta, wd, wp, bias, y, zd, st, B, H, W, C)
            else:
                call("mmd_bifpn_node_fwd_fused", in0, in1, up, pl, theta, wd, wp, bias, sc, sh, y, B, H, W, C)
            e1.record(); torch.cuda.synchronize()
            s = (ctypes.c_ulonglong * 16)()
            assert _lib.LIB.load().mmd_node_stamps(s) == 0
            if it >= 2:
                ev += e0.elapsed_time(e1) * 1e3
                for i in range(7):
                    acc[i] += (s[i + 1] - s[i]) * 0.01
        print(f"{'train' if train else 'eval '} mode {mode}  H {H}  blocks {B * ((H + 7) // 8) ** 2}  event {ev / n:.1f} us  block 0: {sum(acc) / n:.1f} us   " +
              "  ".join(f"{a / n:.2f}" for a in acc))
print("columns: " + " | ".join(names))
