"""Host CPU share of the GPU box vs the thread count torch picks; timing of the tests' CPU work at several thread counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
from helpers import make_state
from oracle import effdet_ref as O
from mm_distillnet_amd.synth import synth_inputs
for nt in (torch.get_num_threads(), 32, 16, 8):
    torch.set_num_threads(nt)
    t = time.time(); spec, st = make_state(2, 8, 24, "audio"); t1 = time.time() - t
    x = synth_inputs(4, 512, seed=1)["audio"]
    t = time.time()
    with torch.no_grad():
        O.forward(st, x, 2, False)
    print("threads %3d: make_state %.2f s, oracle D2 eval forward 4 x 512^2 %.2f s" % (nt, t1, time.time() - t), flush=True)
