"""Weighted micro-benchmark of every 1x1-conv GEMM shape of the bench step (tools/dev/pw_shapes.csv, generated from a
   MMD_PROF_DUMP of bench.py): prints sum(count * time) per family and the worst offenders.  Kernel variants are
   compared by running this under different env toggles inside one gpurun call."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_distillnet_amd import _lib
call = _lib.call
DEV = "cuda:0"
top = int(sys.argv[1]) if len(sys.argv) > 1 else 12
SFX = "_bf16" if os.environ.get("GEMM_BENCH_BF16") else ""      # the bf16 mixed-precision entry points


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


res = []
for line in open(os.path.join(ROOT, "tools/dev/pw_shapes.csv")):
    if line.startswith("#"):
        continue
    M, K, N, f, cnt = (int(v) for v in line.split(","))
    x, w = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) * 0.1
    y = torch.empty(M, N, device=DEV)
    if f < 0:
        dw = torch.zeros(N, K, device=DEV)
        t = timeit(lambda: call("mmd_pwconv_bwd_weight" + SFX, y, x, dw, M, K, N, None, None, 0, None, 1))
    else:
        rpi = 4096 if M % 4096 == 0 else M
        sc = torch.rand(K, device=DEV) + 0.5 if f & 1 else None
        sh = torch.randn(K, device=DEV) if f & 1 else None
        gate = torch.rand(M // rpi, K, device=DEV) if f & 2 else None
        st = torch.zeros(2 * N, dtype=torch.float64, device=DEV) if f & 4 else None
        ws = torch.zeros(64 * 2 * N, dtype=torch.float64, device=DEV) if f & 4 and M >= 16384 else None
        res_ = torch.randn(M, N, device=DEV) if f & 8 else None
        osc = torch.rand(N, device=DEV) if f & 16 else None
        osh = torch.randn(N, device=DEV) if f & 16 else None
        t = timeit(lambda: call("mmd_pwconv_fwd" + SFX, x, w, y, M, K, N, sc, sh, 1 if f & 1 else 0, None, None, None, 0, gate, rpi, None,
                                osc, osh, 0, res_, st, 0, 0, ws, 64 if ws is not None else 0))
    res.append((M, K, N, f, cnt, t))
    del x, w, y
fw = sum(c * t for M, K, N, f, c, t in res if f >= 0) / 1e3
wg = sum(c * t for M, K, N, f, c, t in res if f < 0) / 1e3
print("weighted total: fwd/bwd-data %.3f ms   wgrad %.3f ms" % (fw, wg))
for M, K, N, f, c, t in sorted(res, key=lambda r: -r[4] * r[5])[:top]:
    fl = 2.0 * M * K * N
    print("M%-7d K%-5d N%-5d f%-3d n=%3d  %7.1f us  %6.1f TF  (%.0f us/step)" % (M, K, N, f, c, t, fl / t / 1e6, c * t))
if os.environ.get("GEMM_BENCH_ALL"):
    for r in res:
        print("ALL,%d,%d,%d,%d,%d,%.2f" % r)
