"""LDS-tiled 1x1-conv GEMM: error against float64 and time per shape, for the MFMA form the process environment selects (MMD_SPLIT3 unset:
v_mfma_f32_32x32x2_f32; MMD_SPLIT3=1: six bf16 MFMAs on a three-way split of both operands).  usage: split3_check.py [wide]"""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_distillnet_amd import _lib
call = _lib.call
DEV = "cuda:0"
wide = len(sys.argv) > 1 and sys.argv[1] == "wide"
shapes = [(130944, 112, 112), (43648, 112, 112), (24576, 120, 720), (24576, 720, 120), (24576, 88, 528), (24576, 528, 88), (32768, 288, 48),
          (131072, 24, 144), (393216, 144, 24), (98304, 48, 288)]
junk = torch.empty(64 << 20, device=DEV)
g = torch.Generator(device=DEV); g.manual_seed(1)
print("form:", "split3" if os.environ.get("MMD_SPLIT3") else "fp32 MFMA")
tot = 0.0
for M, K, N in shapes:
    x = torch.randn(M, K, device=DEV, generator=g); w = torch.randn(N, K, device=DEV, generator=g) / math.sqrt(K)
    if wide:      # eight decades of magnitude per operand, signs mixed
        x = x * torch.exp2(torch.randint(-13, 14, (M, K), device=DEV, generator=g).float())
        w = w * torch.exp2(torch.randint(-13, 14, (N, K), device=DEV, generator=g).float())
    y = torch.empty(M, N, device=DEV)
    ts = []
    for it in range(7):
        junk.fill_(1.0); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call("mmd_pwconv_fwd_form", x, w, y, M, K, N, None, None, 0, None, None, None, 0, None, 0, None, None, None, 0, None, None, 0, 0, None, 0, None, 0, 2)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); t = ts[len(ts) // 2]; tot += t
    rows = slice(0, min(M, 16384))
    ref = x[rows].double() @ w.double().t()
    mag = x[rows].double().abs() @ w.double().abs().t()          # sum |a_k b_k|: the scale rounding errors are relative to
    err = (y[rows].double() - ref).abs()
    print(f"M{M} K{K} N{N}: {t:7.1f} us {2.0 * M * K * N / t / 1e6:6.1f} TF   max err / sum|ab| {float((err / mag).max()):.3e}   rms {float((err / mag).pow(2).mean().sqrt()):.3e}")
print(f"sum {tot:.1f} us")
