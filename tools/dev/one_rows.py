"""Time one 1x1-conv GEMM shape (graph of 20 launches, best of 5): python tools/dev/one_rows.py M K N [flags]."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_distillnet_amd import _lib
call = _lib.call
M, K, N = (int(v) for v in sys.argv[1:4])
f = int(sys.argv[4]) if len(sys.argv) > 4 else 0
DEV = "cuda:0"
x, w = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) * 0.1
y = torch.empty(M, N, device=DEV)
rpi = 4096 if M % 4096 == 0 else M
sc = torch.rand(K, device=DEV) + 0.5 if f & 1 else None
sh = torch.randn(K, device=DEV) if f & 1 else None
gate = torch.rand(M // rpi, K, device=DEV) if f & 2 else None
st = torch.zeros(2 * N, dtype=torch.float64, device=DEV) if f & 4 else None
ws = torch.zeros(64 * 2 * N, dtype=torch.float64, device=DEV) if f & 4 and M >= 16384 else None
res_ = torch.randn(M, N, device=DEV) if f & 8 else None
osc = torch.rand(N, device=DEV) if f & 16 else None
osh = torch.randn(N, device=DEV) if f & 16 else None
fn = lambda: call("mmd_pwconv_fwd", x, w, y, M, K, N, sc, sh, 1 if f & 1 else 0, None, None, None, 0, gate, rpi, None, osc, osh, 0, res_, st,
                  0, 0, ws, 64 if ws is not None else 0)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    fn(); fn()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20):
        fn()
g.replay(); torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 20 * 1e6)
print("M%d K%d N%d f%d %s: %.1f us  %.1f TF" % (M, K, N, f, os.environ.get("TAG", ""), best, 2.0 * M * K * N / best / 1e6))
