"""Kernel micro-benchmarks on the GPU box (graph-replayed, so launch overhead of the host is excluded).
   usage: python tools/dev/ubench.py [filter]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_distillnet_amd import _lib
call = _lib.call
DEV = "cuda:0"
flt = sys.argv[1] if len(sys.argv) > 1 else ""


def bench(name, fn, bytes_, flops=0.0, reps=20):
    if flt and flt not in name:
        return
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 5 / reps * 1e6
    print(f"{name:<52} {us:8.1f} us  {bytes_ / us / 1e6:6.2f} TB/s  {flops / us / 1e6:6.1f} TF", flush=True)


def R(*s):
    return torch.randn(*s, device=DEV)


for M, C in ((131072, 144), (32768, 288), (43648, 112), (8192, 720), (524288, 32)):
    x, y = R(M, C), R(M, C)
    sc, sh = torch.rand(C, device=DEV) + 0.5, R(C)
    bench(f"torch copy            M{M} C{C}", lambda: y.copy_(x), 8.0 * M * C)
    bench(f"affine_act swish      M{M} C{C}", lambda: call("mmd_affine_act", x, sc, sh, None, None, None, 0, 1, None, 0, None, y, M, C), 8.0 * M * C)
    mu, istd, ga = R(C) * 0.1, torch.rand(C, device=DEV) + 0.5, torch.rand(C, device=DEV) + 0.5
    sums = torch.zeros(2 * C, dtype=torch.float64, device=DEV)
    dga, dbe = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dz = torch.empty(M, C, device=DEV)
    bench(f"bn_bwd_reduce swish   M{M} C{C}", lambda: call("mmd_bn_bwd_reduce", y, x, sc, sh, mu, istd, 1, None, None, None, 0, None, sums, M, C, None, 0), 8.0 * M * C)
    bench(f"bn_bwd_apply swish    M{M} C{C}", lambda: call("mmd_bn_bwd_apply", y, x, mu, istd, ga, sums, M, dz, dga, dbe, M, C, sc, sh, 1, None, None, None, 0), 12.0 * M * C)

for B, H, C, k, s in ((8, 128, 144, 3, 1), (8, 64, 112, 3, 1), (8, 32, 528, 3, 1), (8, 16, 1248, 3, 1), (8, 32, 112, 3, 1), (8, 256, 32, 3, 1), (8, 256, 16, 3, 1), (8, 64, 288, 5, 1), (8, 32, 720, 5, 1), (8, 256, 96, 3, 2), (8, 16, 1248, 5, 1)):
    x = R(B * H * H, C); w = R(k * k, C)
    OH = -(-H // s)
    y = torch.empty(B * OH * OH, C, device=DEV)
    sc, sh = torch.rand(C, device=DEV) + 0.5, R(C)
    st = torch.zeros(2 * C, dtype=torch.float64, device=DEV)
    by = 4.0 * C * B * (H * H + OH * OH)
    fl = 2.0 * B * OH * OH * C * k * k
    bench(f"dw plain H{H} C{C} k{k} s{s}", lambda: call("mmd_dwconv_fwd", x, w, y, B, H, H, C, k, s, None, None, 0, None, None, None, 0, None, None, 0, None, None, None, 0), by, fl)
    bench(f"dw out-affine+swish only H{H} C{C} k{k} s{s}", lambda: call("mmd_dwconv_fwd", x, w, y, B, H, H, C, k, s, None, None, 0, None, None, None, 0, sc, sh, 1, None, None, None, 0), by, fl)
    bench(f"dw eval  H{H} C{C} k{k} s{s}", lambda: call("mmd_dwconv_fwd", x, w, y, B, H, H, C, k, s, sc, sh, 1, None, None, None, 0, None, None, 0, None, None, None, 0), by, fl)
    bench(f"dw train(stats) H{H} C{C} k{k} s{s}", lambda: call("mmd_dwconv_fwd", x, w, y, B, H, H, C, k, s, sc, sh, 1, None, None, None, 0, None, None, 0, st, None, None, 0), by, fl)
    wsb = torch.zeros(16 * 2 * C, dtype=torch.float64, device=DEV)
    bench(f"dw train(stats slotted) H{H} C{C} k{k} s{s}", lambda: call("mmd_dwconv_fwd", x, w, y, B, H, H, C, k, s, sc, sh, 1, None, None, None, 0, None, None, 0, st, None, wsb, 16), by, fl)

for M, K, N in ((524288, 16, 96), (524288, 32, 16), (131072, 24, 144), (131072, 144, 24), (32768, 48, 288), (32768, 288, 48), (43648, 112, 112),
                (8192, 120, 720), (8192, 720, 120), (8192, 112, 112), (2048, 208, 1248), (2048, 1248, 208), (2048, 352, 2112), (2048, 112, 112), (512, 112, 112)):
    x, w = R(M, K), R(N, K) * 0.1
    y = torch.empty(M, N, device=DEV)
    st = torch.zeros(2 * N, dtype=torch.float64, device=DEV)
    sc, sh = torch.rand(K, device=DEV) + 0.5, R(K)
    by = 4.0 * (M * K + N * K + M * N); fl = 2.0 * M * K * N
    bench(f"pw plain        M{M} K{K} N{N}", lambda: call("mmd_pwconv_fwd", x, w, y, M, K, N, None, None, 0, None, None, None, 0, None, 0, None, None, None, 0, None, None, 0, 0, None, 0), by, fl)
    bench(f"pw in_swish+stats M{M} K{K} N{N}", lambda: call("mmd_pwconv_fwd", x, w, y, M, K, N, sc, sh, 1, None, None, None, 0, None, 0, None, None, None, 0, None, st, 0, 0, None, 0), by, fl)
    wsb = torch.zeros(16 * 2 * N, dtype=torch.float64, device=DEV)
    bench(f"pw in_swish+stats slotted M{M} K{K} N{N}", lambda: call("mmd_pwconv_fwd", x, w, y, M, K, N, sc, sh, 1, None, None, None, 0, None, 0, None, None, None, 0, None, st, 0, 0, wsb, 16), by, fl)
    bench(f"pw stats only M{M} K{K} N{N}", lambda: call("mmd_pwconv_fwd", x, w, y, M, K, N, None, None, 0, None, None, None, 0, None, 0, None, None, None, 0, None, st, 0, 0, None, 0), by, fl)
    bench(f"pw in_swish only M{M} K{K} N{N}", lambda: call("mmd_pwconv_fwd", x, w, y, M, K, N, sc, sh, 1, None, None, None, 0, None, 0, None, None, None, 0, None, None, 0, 0, None, 0), by, fl)
    dw = torch.zeros(N, K, device=DEV)
    bench(f"pw wgrad        M{M} K{K} N{N}", lambda: call("mmd_pwconv_bwd_weight", y, x, dw, M, K, N, None, None, 0, None, 1), by, fl)
