"""What would running the three frozen teachers as ONE batch-24 forward (grouped weights) buy?  Timing-only experiment:
one teacher net fed a 24-image batch stands in for the grouped trio (same kernels, 3x the rows per launch).
   usage: python tools/dev/diag_group.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_distillnet_amd.arch import make_spec
from mm_distillnet_amd.engine import Net
from mm_distillnet_amd.synth import synth_state, synth_inputs

dev = "cuda:0"; S, B = 512, 8
torch.cuda.set_device(0)
tspec, sspec = make_spec(2, 3), make_spec(2, 8)
teachers = [Net(tspec, dev, trainable=False) for _ in range(3)]
for i, t in enumerate(teachers):
    t.load_state(synth_state(tspec, seed=i + 1))
big = Net(tspec, dev, trainable=False)
big.load_state(synth_state(tspec, seed=1))
st = Net(sspec, dev, trainable=True)
st.load_state(synth_state(sspec, seed=9))
x8 = torch.randn(B, 3, S, S, device=dev)
x24 = torch.randn(3 * B, 3, S, S, device=dev)
xa = torch.randn(B, 8, S, S, device=dev)
sides = [torch.cuda.Stream() for _ in range(3)]


def timeit(name, g, n=10):
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    print(f"{name:<52} {(time.perf_counter() - t0) / n * 1e3:8.3f} ms", flush=True)


def cap(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def fwd(net, x, train=False):
    net.begin_step()
    net.forward(x, train=train)


def conc(jobs):
    main = torch.cuda.current_stream()
    ev = main.record_event()
    jobs[0]()
    for side, job in zip(sides, jobs[1:]):
        side.wait_event(ev)
        with torch.cuda.stream(side):
            job()
    for side in sides[:len(jobs) - 1]:
        main.wait_stream(side)


timeit("teacher B=8 alone", cap(lambda: fwd(teachers[0], x8)))
timeit("teacher B=24 alone", cap(lambda: fwd(big, x24)))
timeit("student fwd (train) alone", cap(lambda: fwd(st, xa, True)))
timeit("3 teachers B=8, 3 streams", cap(lambda: conc([lambda t=t: fwd(t, x8) for t in teachers])))
timeit("student + 3 teachers B=8, 4 streams", cap(lambda: conc([lambda: fwd(st, xa, True)] + [lambda t=t: fwd(t, x8) for t in teachers])))
timeit("student + teacher B=24, 2 streams", cap(lambda: conc([lambda: fwd(st, xa, True), lambda: fwd(big, x24)])))
