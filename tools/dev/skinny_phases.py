"""K-loop timeline of the skinny GEMM kernel on a BatchNorm-backward operand launch (block 0's wall-clock stamps; needs a -DMMD_KSTAMPS build:
MMD_EXTRA_HIPCC_FLAGS=-DMMD_KSTAMPS python -m mm_distillnet_amd.build).  usage: skinny_phases.py [M] [Kred] [Nout]"""
import ctypes, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_distillnet_amd import _lib
call = _lib.call
DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
KR = int(sys.argv[2]) if len(sys.argv) > 2 else 1248      # reduction length (the conv's output channels)
NO = int(sys.argv[3]) if len(sys.argv) > 3 else 208       # output width (the conv's input channels)
BK = 128
torch.manual_seed(0)
g, z = torch.randn(M, KR, device=DEV), torch.randn(M, KR, device=DEV)
wt = torch.randn(NO, KR, device=DEV) / math.sqrt(KR)
sc, sh, mu, istd = (torch.rand(KR, device=DEV) + 0.5, torch.randn(KR, device=DEV) * 0.1, torch.randn(KR, device=DEV) * 0.2, torch.rand(KR, device=DEV) + 0.5)
sums = torch.zeros(2 * KR, dtype=torch.float64, device=DEV)
call("mmd_bn_bwd_reduce", g, z, sc, sh, mu, istd, 1, None, None, None, M // 8, None, sums, M, KR, None, 0)
dx = torch.empty(M, NO, device=DEV); dzm = torch.empty(M, KR, device=DEV)
dga = torch.zeros(KR, device=DEV); dbe = torch.zeros(KR, device=DEV)
junk = torch.empty(64 << 20, device=DEV)
nk = (KR + BK - 1) // BK
n = 10
acc = [[0.0] * 5 for _ in range(nk)]
head = tail = tot = ev = 0.0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(n + 2):
    junk.fill_(1.0)
    torch.cuda.synchronize()
    e0.record()
    call("mmd_pwconv_bwd_data_bn", g, z, wt, dx, M, NO, KR, sc, sh, mu, istd, sums, M, 1, None, M // 8, dzm, dga, dbe)
    e1.record()
    torch.cuda.synchronize()
    st = (ctypes.c_ulonglong * 128)()
    assert _lib.LIB.load().mmd_k_stamps(st) == 0
    if it >= 2:
        ev += e0.elapsed_time(e1) * 1e3
        head += (st[1] - st[0]) * 0.01
        for kt in range(nk):
            prev = st[4 * kt] if kt else st[0]
            acc[kt][4] += (st[64 + kt] - prev) * 0.01             # wait for the step's loads
            acc[kt][0] += (st[1 + 4 * kt] - st[64 + kt]) * 0.01   # prologue + LDS store
            acc[kt][1] += (st[2 + 4 * kt] - st[1 + 4 * kt]) * 0.01  # barrier
            acc[kt][2] += (st[3 + 4 * kt] - st[2 + 4 * kt]) * 0.01  # next loads issued + MFMA
            acc[kt][3] += (st[4 + 4 * kt] - st[3 + 4 * kt]) * 0.01  # barrier
        tail += (st[121] - st[120]) * 0.01
        tot += (st[121] - st[0]) * 0.01
print(f"M {M} Kred {KR} Nout {NO}  BK {BK}  steps {nk}   event {ev / n:.1f} us   block 0: {tot / n:.1f} us  (epilogue {tail / n:.1f})")
for kt in range(nk):
    print("   step %2d: load wait %5.2f  prologue+store %5.2f  barrier %5.2f  issue+mma %5.2f  barrier %5.2f" % (kt, acc[kt][4] / n, *(x / n for x in acc[kt][:4])))
