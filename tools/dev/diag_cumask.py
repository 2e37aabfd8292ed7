"""Bounded experiment: the three frozen teachers' forward passes as ONE linear graph on a CU-masked stream
(hipExtStreamCreateWithCUMask) beside the teacher-less step graph (student forward, losses, backward) on the full chip - what a software
pipeline of the teachers one batch ahead could reach when the teachers cannot take CU slots from the backward's latency-bound chain.
   usage: python tools/dev/diag_cumask.py            (B=8, 512x512, D2; GPU box)"""
import ctypes, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as BN
from mm_distillnet_amd.arch import make_spec
from mm_distillnet_amd.synth import synth_inputs
from mm_distillnet_amd.step import DistillEngine, StepConfig

dev = "cuda:0"; S, B = 512, 8
mods = {"rgb": (3, 1), "depth": (3, 2), "thermal": (1, 3)}
specs = {k: make_spec(2, c) for k, (c, _) in mods.items()}
calib = synth_inputs(4, 256, seed=1234)
tstates = {k: BN.calibrated_state(specs[k], seed, calib[k], dev) for k, (_, seed) in mods.items()}
sspec = make_spec(2, 8)
sstate = BN.calibrated_state(sspec, 4, calib["audio"], dev)
batch_cpu = synth_inputs(B, S, seed=24)
for k in tstates:
    BN.tune_teacher_bias(specs[k], tstates[k], batch_cpu[k], dev)
eng = DistillEngine(sspec, specs, dev, StepConfig(image_size=S))
eng.load(sstate, tstates)
batch = {k: v.to(dev) for k, v in batch_cpu.items()}
eng.capture(batch)
ds = eng.static["drop_scale"]
tn = list(eng.teachers.items())
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value)


def cap(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def timeit(name, fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{name:<60} {(time.perf_counter() - t0) / n * 1e3:8.3f} ms", flush=True)


def teachers_serial():
    for mod, net in tn:
        net.begin_step()
        net.forward(eng.static[mod], train=False)


timeit("full step (g_main)", eng.g_main.replay)
g_teach = cap(teachers_serial)
# teacher-less step: the teachers' forwards replaced by their cached outputs
eng.step_body(eng.static, ds)
torch.cuda.synchronize()
cached = {}
for mod, net in tn:
    net.begin_step()
    cached[mod] = net.forward(eng.static[mod], train=False)
    net.begin_step = (lambda: None)
    net.forward = (lambda x, train=False, _m=mod: cached[_m])
torch.cuda.synchronize()
g_noteach = cap(lambda: eng.step_body(eng.static, ds))
timeit("teacher-less step graph alone", g_noteach.replay)
timeit("teacher graph (3 teachers serial) alone, plain stream", g_teach.replay)

masks = {
    "64 CUs: low 64 bits": [0xFFFFFFFF, 0xFFFFFFFF, 0, 0, 0, 0, 0, 0],
    "64 CUs: every 4th bit": [0x11111111] * 8,
    "96 CUs: 3 of every 8 bits": [0x49494949] * 8,
    "128 CUs: every 2nd bit": [0x55555555] * 8,
    "128 CUs: low 128 bits": [0xFFFFFFFF] * 4 + [0] * 4,
}
for name, words in masks.items():
    ms = masked_stream(words)

    def alone():
        ms.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ms):
            g_teach.replay()
        torch.cuda.current_stream().wait_stream(ms)

    def both():
        ms.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ms):
            g_teach.replay()
        g_noteach.replay()
        torch.cuda.current_stream().wait_stream(ms)

    timeit(f"[{name}] teacher graph alone on the masked stream", alone)
    timeit(f"[{name}] teacher graph (masked) || teacher-less step", both)

plain = torch.cuda.Stream()


def both_plain():
    plain.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(plain):
        g_teach.replay()
    g_noteach.replay()
    torch.cuda.current_stream().wait_stream(plain)


timeit("[no mask] teacher graph || teacher-less step", both_plain)
