"""Frozen-net depthwise 3x3/s1 launches (folded BN + swish epilogue, pool): time per launch + a check against torch.
   MMD_DW_ROWS=1 selects the row-streaming kernel (csrc/dw_rows.hip).  usage: python tools/dev/one_dw.py [reps]"""
import math, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mm_distillnet_amd import _lib
call = _lib.call
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def run(x, wd, y, pool, sc, sh, B, H, W, C):
    call("mmd_dwconv_fwd", x, wd, y, B, H, W, C, 3, 1, None, None, 0, None, None, None, 0, sc, sh, 1, None, pool, None, 0)


# correctness on ragged small shapes
for (B, H, W, C) in [(2, 19, 37, 80), (3, 16, 16, 144), (2, 33, 70, 64), (1, 7, 20, 528)]:
    torch.manual_seed(H + W + C)
    x = torch.randn(B, C, H, W); wd = torch.randn(C, 1, 3, 3) / 3
    sc, sh = torch.rand(C) + 0.5, torch.randn(C) * 0.2
    ref = F.conv2d(x, wd, padding=1, groups=C) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    ref = ref * torch.sigmoid(ref)
    xn = x.permute(0, 2, 3, 1).contiguous().cuda(); wn = wd.reshape(C, 9).t().contiguous().cuda()
    y = torch.full((B * H * W, C), float("nan"), device="cuda"); pool = torch.zeros(B, C, device="cuda")
    run(xn, wn, y, pool, sc.cuda(), sh.cuda(), B, H, W, C)
    err = (y.view(B, H, W, C).cpu() - ref.permute(0, 2, 3, 1)).abs().max().item()
    perr = (pool.cpu() - ref.mean((2, 3))).abs().max().item()
    print(f"check B{B} H{H} W{W} C{C}: max|err| {err:.2e} pool {perr:.2e}", flush=True)
    assert err < 1e-4 and perr < 1e-4

B = 8
for H, C in [(128, 144), (64, 112), (32, 528), (32, 112), (16, 2112), (16, 1248), (256, 64), (64, 288)]:
    x = torch.randn(B * H * H, C, device="cuda"); wd = torch.randn(9, C, device="cuda") / 3
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.2
    y = torch.empty_like(x); pool = torch.zeros(B, C, device="cuda")
    for _ in range(3):
        run(x, wd, y, pool, sc, sh, B, H, H, C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run(x, wd, y, pool, sc, sh, B, H, H, C)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e3 / reps
    print(f"H{H:4d} C{C:5d}: {t:7.1f} us  {8.0 * x.numel() / t * 1e-6:5.2f} TB/s", flush=True)
