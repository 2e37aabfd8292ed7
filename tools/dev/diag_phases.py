"""Phase timing on the GPU box: each phase of the distillation step captured as its own hipGraph and replayed.
   usage: python tools/dev/diag_phases.py            (B=8, 512x512, D2)"""
import os, sys, time, torch
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
for _n in [eng.student] + list(eng.teachers.values()):      # the phase graphs below allocate beyond what the step's graph froze
    _n.arena.frozen = False; _n.zarena.frozen = False
eng.ws.frozen = False


def timeit(name, g, n=10):
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    print(f"{name:<44} {(time.perf_counter() - t0) / n * 1e3:8.3f} ms", flush=True)


def cap(fn, stream=None):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with (torch.cuda.graph(g) if stream is None else torch.cuda.graph(g, stream=stream)):
        fn()
    return g


timeit("full step (g_main)", eng.g_main)
timeit("optimizer (g_opt)", eng.g_opt)
st = eng.student
ds = eng.static["drop_scale"]


def student_fwd():
    st.begin_step()
    st.forward(eng.static["audio"], train=True, drop_scale=ds)


timeit("student forward (train)", cap(student_fwd))
tn = list(eng.teachers.items())


def one_teacher():
    mod, net = tn[0]
    net.begin_step()
    c, r, f = net.forward(eng.static[mod], train=False)


timeit("one teacher forward", cap(one_teacher))


def one_teacher_pl():
    mod, net = tn[0]
    eng.ws.reset()
    eng.mask_ws = eng.ws.alloc((B * 1024 * 16,), torch.int64)
    net.begin_step()
    c, r, f = net.forward(eng.static[mod], train=False)
    eng._pseudo_labels(net, c, r, B, c.shape[1], S)


timeit("one teacher forward + pseudo labels", cap(one_teacher_pl))


def teachers_serial():
    for mod, net in tn:
        net.begin_step()
        net.forward(eng.static[mod], train=False)


timeit("3 teachers, one stream", cap(teachers_serial))


def teachers_conc():
    main = torch.cuda.current_stream()
    ev = main.record_event()
    for i, (mod, net) in enumerate(tn):
        side = eng.side_streams[i]
        side.wait_event(ev)
        with torch.cuda.stream(side):
            net.begin_step()
            net.forward(eng.static[mod], train=False)
    for side in eng.side_streams:
        main.wait_stream(side)


timeit("3 teachers, side streams", cap(teachers_conc))


def all_fwd():
    main = torch.cuda.current_stream()
    ev = main.record_event()
    student_fwd()
    for i, (mod, net) in enumerate(tn):
        side = eng.side_streams[i]
        side.wait_event(ev)
        with torch.cuda.stream(side):
            net.begin_step()
            net.forward(eng.static[mod], train=False)
    for side in eng.side_streams:
        main.wait_stream(side)


timeit("student + 3 teachers forward, 4 streams", cap(all_fwd))


def fwd_k(k):
    def f():
        main = torch.cuda.current_stream()
        ev = main.record_event()
        student_fwd()
        for i, (mod, net) in enumerate(tn[:k]):
            side = eng.side_streams[i]
            side.wait_event(ev)
            with torch.cuda.stream(side):
                net.begin_step()
                net.forward(eng.static[mod], train=False)
        for side in eng.side_streams[:k]:
            main.wait_stream(side)
    return f


# is the forward phase bound by the student's chain (flat in k) or by the chip's throughput (linear in k)?
for k in (1, 2):
    timeit("student + %d teacher(s) forward" % k, cap(fwd_k(k)))
# proxy for a cross-teacher BATCHED pack (one launch per layer over 3 x B images, per-group weights): the same kernels on one frozen net at
# batch 3B - same work and launch count as a pack would have, weights shared instead of per group
if os.environ.get("MMD_DIAG_PACK"):
    from mm_distillnet_amd.engine import Net
    net24 = Net(specs["rgb"], dev, trainable=False)
    net24.load_state(tstates["rgb"])
    x24 = torch.cat([eng.static["rgb"]] * 3, 0).contiguous()

    def pack_fwd():
        net24.begin_step()
        net24.forward(x24, train=False)

    timeit("one frozen net at batch 3B (pack proxy)", cap(pack_fwd))

    def student_and_pack():
        main = torch.cuda.current_stream()
        ev = main.record_event()
        student_fwd()
        side = eng.side_streams[0]
        side.wait_event(ev)
        with torch.cuda.stream(side):
            pack_fwd()
        main.wait_stream(side)

    timeit("student + frozen net at batch 3B, 2 streams", cap(student_and_pack))
if getattr(eng, "pack", False):
    def real_pack():
        nets = [n for _, n in tn]
        nets[0].begin_step()
        nets[0].forward([eng.static[m] for m, _ in tn], train=False, pack=nets)

    timeit("teacher pack (3 nets, one launch per layer)", cap(real_pack))

    def student_and_real_pack():
        main = torch.cuda.current_stream()
        ev = main.record_event()
        student_fwd()
        side = eng.side_streams[0]
        side.wait_event(ev)
        with torch.cuda.stream(side):
            real_pack()
        main.wait_stream(side)

    timeit("student + teacher pack, 2 streams", cap(student_and_real_pack))
    if os.environ.get("MMD_DIAG_HALF"):
        # the pack as two half batches (3 x 4 images each): do the smaller activations buy cache hits worth their doubled launch count?
        def half_packs():
            nets = [n for _, n in tn]
            nets[0].begin_step()
            for lo in (0, B // 2):
                nets[0].forward([eng.static[m][lo:lo + B // 2].contiguous() for m, _ in tn], train=False, pack=nets)

        timeit("teacher pack as two half batches", cap(half_packs))

        def student_and_half_packs():
            main = torch.cuda.current_stream()
            ev = main.record_event()
            student_fwd()
            side = eng.side_streams[0]
            side.wait_event(ev)
            with torch.cuda.stream(side):
                half_packs()
            main.wait_stream(side)

        timeit("student + two half packs, 2 streams", cap(student_and_half_packs))
if os.environ.get("MMD_DIAG_FWD_ONLY"):
    sys.exit(0)

# student-only step: teachers replaced by their cached outputs
eng.step_body(eng.static, ds)
torch.cuda.synchronize()
cached = {}
use_pack = getattr(eng, "pack", False)
if use_pack:
    _nets = [n for _, n in tn]
    _nets[0].begin_step()
    cached["pack"] = _nets[0].forward([eng.static[m] for m, _ in tn], train=False, pack=_nets)
for mod, net in tn:
    if not use_pack:
        net.begin_step()
        cached[mod] = net.forward(eng.static[mod], train=False)
    net.begin_step = (lambda: None)
    net.forward = (lambda x, train=False, pack=None, _m=mod: cached["pack"] if pack is not None else cached[_m])
torch.cuda.synchronize()
timeit("step without teacher forwards", cap(lambda: eng.step_body(eng.static, ds)))

# ---- what would software-pipelining the frozen teachers one batch ahead buy?  (teacher graph on a second stream, replayed
# together with the teacher-less step graph)
g_noteach = cap(lambda: eng.step_body(eng.static, ds))
for mod, net in tn:              # restore the real forwards for the teacher graph
    del net.begin_step, net.forward
g_teach = cap(real_pack if use_pack else teachers_conc)
s2 = torch.cuda.Stream()
def both():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        g_teach.replay()
    g_noteach.replay()
    torch.cuda.current_stream().wait_stream(s2)
for _ in range(3):
    both()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    both()
torch.cuda.synchronize()
print(f"{'teacher graph || teacher-less step graph':<44} {(time.perf_counter() - t0) / 10 * 1e3:8.3f} ms", flush=True)
timeit("teacher-less step graph alone", g_noteach)
timeit("teacher graph alone", g_teach)

# ---- the same pair with the teacher-less step graph replayed on (and, second variant, captured on) a HIGH-priority stream: do the
# student's kernels then get the CUs first?
s_hi = torch.cuda.Stream(priority=-1)
def both_prio(g_student):
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur); s_hi.wait_stream(cur)
    with torch.cuda.stream(s2):
        g_teach.replay()
    with torch.cuda.stream(s_hi):
        g_student.replay()
    cur.wait_stream(s2); cur.wait_stream(s_hi)
def time_pair(label, g_student):
    for _ in range(3):
        both_prio(g_student)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        both_prio(g_student)
    torch.cuda.synchronize()
    print(f"{label:<44} {(time.perf_counter() - t0) / 10 * 1e3:8.3f} ms", flush=True)
time_pair("teacher graph || step graph on a high-priority stream", g_noteach)
for mod, net in tn:
    net.begin_step = (lambda: None)
    net.forward = (lambda x, train=False, pack=None, _m=mod: cached["pack"] if pack is not None else cached[_m])
g_hi = cap(lambda: eng.step_body(eng.static, ds), stream=s_hi)
torch.cuda.synchronize()
time_pair("  ... and captured on it", g_hi)
