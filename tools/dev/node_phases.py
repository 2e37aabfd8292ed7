"""Phase timeline of the whole-node BiFPN backward kernel on a small map (block 0's wall-clock stamps; needs a -DMMD_NODE_TIMING build:
MMD_EXTRA_HIPCC_FLAGS=-DMMD_NODE_TIMING python -m mm_distillnet_amd.build).  usage: node_phases.py [H] [C]"""
import ctypes, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_distillnet_amd import _lib
call = _lib.call
DEV = "cuda:0"
H = int(sys.argv[1]) if len(sys.argv) > 1 else 8
C = int(sys.argv[2]) if len(sys.argv) > 2 else 112
B, W = 8, H
g = lambda t: t.to(DEV).contiguous()
names = ["entry->coef barrier", "g,z tile staged", "MFMA + dzd tile", "depthwise^T", "operand pass (loads, stores, atomics)", "up-sample sums", "barrier", "atomics tail"]
for mode in ("td", "bu", "p7"):
    M = B * H * W
    has1, hasu, hasp = mode == "bu", mode == "td", mode in ("bu", "p7")
    torch.manual_seed(0)
    in0 = g(torch.randn(M, C)); in1 = g(torch.randn(M, C)) if has1 else None
    up = g(torch.randn(M // 4, C)) if hasu else None
    pl = g(torch.randn(4 * M, C) - 1.0) if hasp else None
    theta = g(torch.tensor([0.7, 1.3, 0.4][:2 if mode != "bu" else 3]))
    wd = g(torch.randn(9, C) / 3); wp = g(torch.randn(C, C) / math.sqrt(C))      # (the transposed weight: any [C, C] matrix times alike)
    gg, zz = g(torch.randn(M, C)), g(torch.randn(M, C) * 1.2 + 0.1)
    sc, sh, mu, istd = (g(t) for t in (torch.rand(C) + 0.5, torch.randn(C) * 0.1, torch.randn(C) * 0.2, torch.rand(C) + 0.5))
    sums = torch.zeros(2 * C, dtype=torch.float64, device=DEV)
    call("mmd_bn_bwd_reduce", gg, zz, sc, sh, mu, istd, 0, None, None, None, H * W, None, sums, M, C, None, 0)
    wdot, dwg, e0 = torch.zeros(4, device=DEV), torch.zeros(9, C, device=DEV), torch.zeros(M, C, device=DEV)
    e1 = torch.zeros(M, C, device=DEV) if has1 else None
    eu = torch.zeros(M // 4, C, device=DEV) if hasu else None
    ep = torch.zeros(4 * M, C, device=DEV) if hasp else None
    dzm = torch.empty(M, C, device=DEV); dga = torch.zeros(C, device=DEV); dbe = torch.zeros(C, device=DEV)
    none12 = (None,) * 12
    junk = torch.empty(64 << 20, device=DEV)
    acc = [0.0] * 8
    sub = [0.0] * 4
    n = 20
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for it in range(n + 2):
        junk.fill_(1.0)            # evict L2 so operands come from HBM as in the step
        torch.cuda.synchronize()
        ev0.record()
        call("mmd_bifpn_node_bwd_full", in0, in1, up, pl, theta, wd, wdot, B, H, W, C, e0, 0, e1, 0, eu, 0, dwg, *none12, ep, None, None, None, None, 0,
             None, None, gg, zz, sc, mu, istd, sums, M, wp, dzm, dga, dbe)
        ev1.record()
        torch.cuda.synchronize()
        st = (ctypes.c_ulonglong * 16)()
        rc = _lib.LIB.load().mmd_node_stamps(st)
        assert rc == 0
        if it >= 2:
            tot += ev0.elapsed_time(ev1) * 1e3
            for i in range(8):
                acc[i] += (st[i + 1] - st[i]) * 0.01      # 100 MHz -> us
            for i, (x, y) in enumerate(((4, 9), (9, 10), (10, 11), (11, 12))):
                sub[i] += (st[y] - st[x]) * 0.01
    print(f"mode {mode}  H {H}  C {C}  blocks {B * ((H + 7) // 8) ** 2 * ((C + 63) // 64)}   event time {tot / n:.1f} us   block 0: {sum(acc) / n:.1f} us")
    for i in range(8):
        print(f"    {names[i]:<40} {acc[i] / n:6.2f} us")
    print("    operand pass: pair 0 loads %.2f, consume %.2f, pair 1 loads %.2f, consume %.2f us (16-channel form: one round, the second pair is stale)" % tuple(x / n for x in sub))
