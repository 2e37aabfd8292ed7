"""Weighted micro-benchmark of the weight-gradient kernels of one bench step (they run on their own stream, but
`MMD_DEV_SKIP_WG=1 python bench.py` shows they cost the step ~4.6 ms of its 22.7: they do not hide behind the main chain).
   1x1 shapes: tools/dev/pw_shapes.csv rows with flags = -1; depthwise shapes: the dwwg tags of a MMD_PROF_DUMP."""
import ctypes, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_distillnet_amd import _lib
call = _lib.call
DEV = "cuda:0"
DW = [(64, 112, 3, 1, 5), (32, 112, 3, 1, 10), (32, 720, 5, 1, 3), (256, 16, 3, 1, 1), (64, 288, 5, 1, 2), (128, 144, 3, 1, 2),
      (16, 112, 3, 1, 10), (16, 1248, 5, 1, 4), (8, 112, 3, 1, 10), (256, 96, 3, 2, 1), (256, 32, 3, 1, 1), (128, 144, 5, 2, 1),
      (32, 528, 3, 1, 3), (4, 112, 3, 1, 5), (32, 528, 5, 1, 1), (32, 720, 5, 2, 1), (64, 288, 3, 2, 1), (16, 2112, 3, 1, 1),
      (16, 1248, 3, 1, 1)]


def timeit(fn, reps=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e6)
    return best


rows = []
B = 8
for H, C, k, s, cnt in DW:
    OH = -(-H // s)
    x = torch.randn(B * H * H, C, device=DEV); dy = torch.randn(B * OH * OH, C, device=DEV)
    dw = torch.zeros(k * k, C, device=DEV)
    sc, sh = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
    t = timeit(lambda: call("mmd_dwconv_bwd_weight", x, dy, dw, B, H, H, C, k, s, sc, sh, 1))
    rows.append(("dw H%d C%d k%d s%d" % (H, C, k, s), cnt, t, 4.0 * (x.numel() + dy.numel())))
sizes = [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)]
flat = [5, B] + [v for hw in sizes for v in hw]
desc = (ctypes.c_int * len(flat))(*flat)
Mt = sum((B * h * w + 127) // 128 * 128 for h, w in sizes)
x = torch.randn(Mt, 112, device=DEV); dy = torch.randn(Mt, 112, device=DEV); dw = torch.zeros(9, 112, device=DEV)
t = timeit(lambda: call("mmd_dwconv3_pyr_bwd_weight", x, dy, dw, desc, 112, None, None, 0, 0))
rows.append(("dwpyr C112", 8, t, 8.0 * x.numel()))
for line in ([] if os.environ.get("WG_DW_ONLY") else open(os.path.join(ROOT, "tools/dev/pw_shapes.csv"))):
    if line.startswith("#"):
        continue
    M, K, N, f, cnt = (int(v) for v in line.split(","))
    if f >= 0:
        continue
    x, y = torch.randn(M, K, device=DEV), torch.randn(M, N, device=DEV)
    dwt = torch.zeros(N, K, device=DEV)
    if os.environ.get("WG_BNP"):      # the in-step form: BatchNorm backward evaluated on the dY operand (90 of 99 launches)
        z = torch.randn(M, N, device=DEV)
        sc, sh = torch.rand(N, device=DEV) + 0.5, torch.randn(N, device=DEV) * 0.1
        mu, istd = torch.randn(N, device=DEV) * 0.1, torch.rand(N, device=DEV) + 0.5
        sums = torch.randn(2 * N, dtype=torch.float64, device=DEV)
        dga, dbe = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
        act = 1 if os.environ["WG_BNP"] == "2" else 0
        t = timeit(lambda: call("mmd_pwconv_bwd_weight_bn", y, z, x, dwt, M, K, N, None, None, 0, None, 1, sc, sh, mu, istd, sums, M, act,
                                None, 1, dga, dbe))
    else:
        t = timeit(lambda: call("mmd_pwconv_bwd_weight", y, x, dwt, M, K, N, None, None, 0, None, 1))
    rows.append(("pw M%d K%d N%d" % (M, K, N), cnt, t, 4.0 * M * (K + N)))
    del x, y
print("weighted: dw %.3f ms   pw %.3f ms" % (sum(c * t for n, c, t, b in rows if n.startswith("dw")) / 1e3,
                                              sum(c * t for n, c, t, b in rows if n.startswith("pw")) / 1e3))
for n, c, t, b in sorted(rows, key=lambda r: -r[1] * r[2])[: int(sys.argv[1]) if len(sys.argv) > 1 else 30]:
    print("%-28s n=%3d %7.1f us  %5.2f TB/s  (%.0f us/step)" % (n, c, t, b / t / 1e6, c * t))
