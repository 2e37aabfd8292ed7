"""Time the fused expand+depthwise kernel against the two-kernel path on the frozen-net front blocks of D2 at 512^2, B = 8.
usage: python tools/dev/one_mbx.py [reps]   (MMD_MBX_WAVES=3|6 picks the block size)"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mm_distillnet_amd import _lib

call = _lib.call
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 8
SHAPES = [(16, 96, 3, 2, 256), (24, 144, 3, 1, 128), (24, 144, 5, 2, 128), (48, 288, 5, 1, 64), (48, 288, 3, 2, 64)]
if os.environ.get("MBX_D4"):
    SHAPES = [(24, 144, 3, 2, 384), (32, 192, 3, 1, 192), (32, 192, 5, 2, 192), (56, 336, 5, 1, 96), (56, 336, 3, 2, 96)]


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for cin, cmid, k, s, H in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(B * H * H, cin, device="cuda")
    w0 = torch.randn(cmid, cin, device="cuda") / math.sqrt(cin)
    sc0, sh0 = torch.rand(cmid, device="cuda") + 0.5, torch.randn(cmid, device="cuda") * 0.2
    wd = torch.randn(k * k, cmid, device="cuda") / k
    sc1, sh1 = torch.rand(cmid, device="cuda") + 0.5, torch.randn(cmid, device="cuda") * 0.2
    OH = -(-H // s)
    y = torch.empty(B * OH * OH, cmid, device="cuda")
    y2 = torch.empty_like(y)
    ez = torch.empty(B * H * H, cmid, device="cuda")
    pool = None if os.environ.get("MBX_NOPOOL") else torch.zeros(B, cmid, device="cuda")

    def fused():
        call("mmd_mbconv_expand_dw_fwd", x, w0, sc0, sh0, wd, sc1, sh1, y, pool, B, H, H, cin, cmid, k, s)

    def pw():
        call("mmd_pwconv_fwd", x, w0, ez, B * H * H, cin, cmid, None, None, 0, None, None, None, 0, None, H * H, None, sc0, sh0, 1,
             None, None, 0, 0, None, 0)

    def dw():
        call("mmd_dwconv_fwd", ez, wd, y2, B, H, H, cmid, k, s, None, None, 0, None, None, None, 0, sc1, sh1, 1, None, pool, None, 0)

    tf, tp, td = timed(fused), timed(pw), timed(dw)
    err = (y - y2).abs().max().item()
    by = 4.0 * (B * H * H * cin + B * OH * OH * cmid)
    print(f"Cin{cin:3d} C{cmid:4d} k{k} s{s} H{H:4d}: fused {tf:7.1f} us ({by / tf * 1e-6:5.2f} TB/s)   pw {tp:6.1f} + dw {td:6.1f} = {tp + td:7.1f} us"
          f"   x{(tp + td) / tf:4.2f}   max|diff| {err:.2e}", flush=True)
