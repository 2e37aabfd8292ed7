"""Run one 1x1-conv GEMM shape a few times (for rocprofv3 --pmc / --kernel-trace): python3 one_gemm.py M K N [reps]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_distillnet_amd import _lib
M, K, N = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
x, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * 0.1
y = torch.empty(M, N, device="cuda")
for _ in range(reps):
    _lib.call("mmd_pwconv_fwd", x, w, y, M, K, N, None, None, 0, None, None, None, 0, None, 0, None, None, None, 0, None, None, 0, 0, None, 0)
torch.cuda.synchronize()
