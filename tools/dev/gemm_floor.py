import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_distillnet_amd import _lib
call = _lib.call
def timeit(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
for M, K, N in ((128, 32, 64), (8192, 32, 64), (32768, 32, 64), (98304, 32, 64), (32768, 64, 64), (32768, 128, 64), (32768, 256, 64), (32768, 512, 64), (32768, 32, 128), (32768, 32, 256)):
    x, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * 0.1
    y = torch.empty(M, N, device="cuda")
    t = timeit(lambda: call("mmd_pwconv_fwd", x, w, y, M, K, N, None, None, 0, None, None, None, 0, None, 0, None, None, None, 0, None, None, 0, 0, None, 0))
    print("tiles=%d ktiles=%d M%d K%d N%d: %.1f us" % (-(-M // 128) * -(-N // 64), -(-K // 32), M, K, N, t), flush=True)
