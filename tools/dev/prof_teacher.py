"""One frozen teacher forward (D2, 512^2, B = 8) run N times eagerly - meant to sit under `rocprofv3 --kernel-trace --stats`:
   rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/teacher -o t -- python3 tools/dev/prof_teacher.py 10
Prints the wall time per forward as well (graph replay)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_distillnet_amd.arch import make_spec
from mm_distillnet_amd.engine import Net
from mm_distillnet_amd.synth import synth_state

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev, S, B = "cuda:0", 512, 8
spec = make_spec(2, 3)
net = Net(spec, dev, trainable=False)
net.load_state(synth_state(spec, seed=1))
x = torch.randn(B, 3, S, S, device=dev)


def fwd():
    net.begin_step()
    net.forward(x, train=False)


for _ in range(2):
    fwd()
torch.cuda.synchronize()
for _ in range(n):
    fwd()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fwd()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print(f"teacher forward, graph replay: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
