"""Thin 1x1 convs of the 256^2 / 128^2 stages: the LDS-tiled kernels (form 2) against the thin-K row-slab kernel (form 1) per shape, frozen-net
epilogue (gate, folded BN, residual).  usage: thin_forms.py"""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_distillnet_amd import _lib
call = _lib.call
DEV = "cuda:0"
shapes = [(1572864, 16, 16, True), (1572864, 32, 16, False), (524288, 16, 16, True), (524288, 32, 16, False), (393216, 144, 24, True),
          (131072, 24, 144, False), (98304, 288, 48, True), (393216, 96, 24, False)]
junk = torch.empty(64 << 20, device=DEV)
for M, K, N, res in shapes:
    B = 24 if M % 24 == 0 else 8
    rpi = M // B
    x = torch.randn(M, K, device=DEV); w = torch.randn(N, K, device=DEV) / math.sqrt(K)
    gate = torch.rand(B, K, device=DEV); osc = torch.rand(N, device=DEV) + 0.5; osh = torch.randn(N, device=DEV) * 0.1
    r = torch.randn(M, N, device=DEV) if res else None
    y = torch.empty(M, N, device=DEV)
    out = []
    for form in (2, 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for it in range(7):
            junk.fill_(1.0); torch.cuda.synchronize()
            e0.record()
            call("mmd_pwconv_fwd_form", x, w, y, M, K, N, None, None, 0, None, None, None, 0, gate, rpi, None, osc, osh, 0, r, None, 0, 0, None, 0, None, 0, form)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort(); out.append(ts[len(ts) // 2])
    by = 4.0 * (M * K + M * N * (2 if res else 1))
    print(f"M{M} K{K} N{N} res{int(res)}: tiled {out[0]:7.1f} us ({by / out[0] / 1e6:5.2f} TB/s)   rows {out[1]:7.1f} us ({by / out[1] / 1e6:5.2f} TB/s)")
