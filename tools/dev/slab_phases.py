"""Timeline of the slab GEMM kernel (csrc/pw_slab.hip) on a BatchNorm-backward operand launch: block 0's wall-clock stamps.  Needs a
-DMMD_SLSTAMPS build of the library (MMD_LIB=<that .so>).  usage: slab_phases.py [M] [Kred] [Nout]"""
import ctypes, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_distillnet_amd import _lib
call = _lib.call
DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
KR = int(sys.argv[2]) if len(sys.argv) > 2 else 1248      # reduction length (the conv's output channels)
NO = int(sys.argv[3]) if len(sys.argv) > 3 else 208       # output width (the conv's input channels)
torch.manual_seed(0)
g, z = torch.randn(M, KR, device=DEV), torch.randn(M, KR, device=DEV)
wt = torch.randn(NO, KR, device=DEV) / math.sqrt(KR)
sc, sh, mu, istd = (torch.rand(KR, device=DEV) + 0.5, torch.randn(KR, device=DEV) * 0.1, torch.randn(KR, device=DEV) * 0.2, torch.rand(KR, device=DEV) + 0.5)
sums = torch.zeros(2 * KR, dtype=torch.float64, device=DEV)
call("mmd_bn_bwd_reduce", g, z, sc, sh, mu, istd, 1, None, None, None, M // 8, None, sums, M, KR, None, 0)
dx = torch.zeros(M, NO, device=DEV); dzm = torch.empty(M, KR, device=DEV)
dga = torch.zeros(KR, device=DEV); dbe = torch.zeros(KR, device=DEV)
xs = torch.zeros(2 * NO, dtype=torch.float64, device=DEV)
zup, mup, iup = torch.randn(M, NO, device=DEV), torch.randn(NO, device=DEV), torch.rand(NO, device=DEV) + 0.5
dll = _lib.LIB.load()
nws = int(dll.mmd_pwconv_slab_ws_floats(M, KR, NO, 1))
ws = torch.empty(max(nws, 1), device=DEV)
junk = torch.empty(64 << 20, device=DEV)
n = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for form in (4, 2):
    ev, rows = 0.0, None
    for it in range(n + 2):
        junk.fill_(1.0)
        torch.cuda.synchronize()
        e0.record()
        call("mmd_pwconv_bwd_data_bn2_form", g, z, wt, dx, M, NO, KR, sc, sh, mu, istd, sums, M, 1, None, M // 8, dzm, dga, dbe, dx,
             zup, mup, iup, None, M // 8, xs, None, 0, None, None, None, None, None, None, 0, ws if nws else None, nws, form)
        e1.record()
        torch.cuda.synchronize()
        if it >= 2:
            ev += e0.elapsed_time(e1) * 1e3
            if form == 4:
                st = (ctypes.c_ulonglong * 128)()
                assert dll.mmd_slab_stamps(st) == 0
                v = [x * 0.01 for x in st]
                rows = v if rows is None else [a + b for a, b in zip(rows, v)]
    print(f"form {form}: M {M} Kred {KR} Nout {NO}  ws floats {nws}  event {ev / n:.1f} us")
    if rows:
        r = [x / n for x in rows]
        print(f"   block 0 / wave 0: start -> first loads issued + table filled {r[1] - r[0]:.2f}")
        prev, i = r[1], 0
        while 2 + i < 60 and r[2 + i] > prev:
            print(f"   granule {i}: {r[2 + i] - prev:5.2f}")
            prev = r[2 + i]; i += 1
        print(f"   last granule: {r[60] - prev:5.2f}")
        print(f"   reduction + tile {r[61] - r[60]:.2f}   write-out / epilogue {r[62] - r[61]:.2f}   kernel body {r[62] - r[0]:.2f}")
        if r[65] > r[64] > 0:
            print(f"   combine launch: starts {r[64] - r[62]:.2f} after block 0's end, body {r[65] - r[64]:.2f};  first stamp -> last stamp {r[65] - r[0]:.2f}")
