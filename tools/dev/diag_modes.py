"""Is the occasional fast run (18.4 vs 19.3 ms/step) a property of the process or of one engine instance?  Builds the engine several
times in ONE process (new allocations, new graph capture each time) and times 30 replays of each.
   usage: python tools/dev/diag_modes.py [instances]"""
import gc, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as BN
from mm_distillnet_amd.arch import make_spec
from mm_distillnet_amd.synth import synth_inputs
from mm_distillnet_amd.step import DistillEngine, StepConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
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
batch = {k: v.to(dev) for k, v in batch_cpu.items()}
pad = []
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
raw = []
NEW = [int(v) for v in os.environ.get("MMD_DIAG_NEWSTREAMS", "").split(",") if v] or [0] * n      # raw hipStreamCreate calls before instance i
for i in range(n):
    for _ in range(NEW[i] if i < len(NEW) else 0):
        h = ctypes.c_void_p()
        hip.hipStreamCreate(ctypes.byref(h)); raw.append(h)
        scratch = torch.zeros(64, device=dev)
        hip.hipMemsetAsync(ctypes.c_void_p(scratch.data_ptr()), 0, ctypes.c_size_t(256), h)      # first use: binds the stream to a hardware queue
        hip.hipStreamSynchronize(h)
    if os.environ.get("MMD_DIAG_PAD"):      # shift every later allocation by an odd amount
        pad.append(torch.empty((i + 1) * int(os.environ["MMD_DIAG_PAD"]), dtype=torch.uint8, device=dev))
    eng = DistillEngine(sspec, specs, dev, StepConfig(image_size=S))
    eng.load(sstate, tstates)
    eng.capture(batch)
    for _ in range(5):
        eng.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        eng.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 30 * 1e3
    ptrs = [eng.student.ps.flat.data_ptr(), eng.ws.chunks[0].data_ptr() if hasattr(eng.ws, "chunks") and eng.ws.chunks else 0]
    print(f"instance {i}: {ms:.3f} ms/step   flat@{ptrs[0] % (1 << 21):#x} ws@{ptrs[1] % (1 << 21):#x}", flush=True)
    del eng
    gc.collect(); torch.cuda.empty_cache()
