"""Per-kernel cost of a dependent chain inside a hipGraph: tiny kernels of this library and torch."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_distillnet_amd import _lib
call = _lib.call
def timeit(fn, reps=50):
    fn(); fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e6)
    return best
x = torch.zeros(256, device="cuda"); y = torch.zeros(256, device="cuda")
th = torch.ones(2, device="cuda"); wd = torch.zeros(4, device="cuda"); dth = torch.zeros(2, device="cuda")
print("torch add_ 256 elems      %.2f us" % timeit(lambda: x.add_(1.0)))
print("mmd_scale_acc 256 elems   %.2f us" % timeit(lambda: call("mmd_scale_acc", x, y, None, 0, 0, 1, 256)))
print("mmd_bifpn_theta_bwd       %.2f us" % timeit(lambda: call("mmd_bifpn_theta_bwd", th, wd, dth, 2)))
big = torch.zeros(1 << 22, device="cuda"); big2 = torch.zeros(1 << 22, device="cuda")
print("mmd_scale_acc 16 MB       %.2f us" % timeit(lambda: call("mmd_scale_acc", big, big2, None, 0, 0, 1, big.numel())))
a, b = torch.zeros(256, device="cuda"), torch.zeros(256, device="cuda")
def two():
    call("mmd_scale_acc", x, y, None, 0, 0, 1, 256); call("mmd_scale_acc", a, b, None, 0, 0, 1, 256)
print("2 independent tiny (same stream) per kernel %.2f us" % (timeit(two) / 2))
