"""Where the host time of DeviceInputPipeline.submit() goes (GPU box): the stages of one submit timed alone on an idle GPU,
then the whole submit while a captured step graph replays back to back."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mm_distillnet_amd.data import RawSyntheticMultimodalDetection, DeviceInputPipeline

dev = "cuda:0"
ds = RawSyntheticMultimodalDetection({"synthetic_length": 64}, "train")
batches = [[ds[i * 8 + k] for k in range(8)] for i in range(4)]
pipe = DeviceInputPipeline(512, dev)
for i in range(3):
    pipe.submit(batches[i % 4]).wait()
torch.cuda.synchronize()

def timed(n, f):
    t = time.perf_counter()
    for i in range(n):
        f(i)
    return (time.perf_counter() - t) / n * 1e3

print("submit + wait + device sync, idle GPU: %.2f ms" % timed(20, lambda i: (pipe.submit(batches[i % 4]).wait(), torch.cuda.synchronize())))
print("submit only (no device sync), idle GPU: %.2f ms" % timed(20, lambda i: pipe.submit(batches[i % 4])))
torch.cuda.synchronize()
# the pieces
smp = batches[0]
t = time.perf_counter()
for _ in range(20):
    for b, s in enumerate(smp):
        for k in ("rgb", "depth", "thermal", "audio"):
            pipe._pinned[(k, b)].copy_(s[k])
print("32 pinned copy_: %.2f ms" % ((time.perf_counter() - t) / 20 * 1e3))
t = time.perf_counter()
with torch.cuda.stream(pipe.stream):
    for _ in range(20):
        for b, s in enumerate(smp):
            for k in ("rgb", "depth", "thermal", "audio"):
                pipe._pinned[(k, b)].to(dev, non_blocking=True)
print("32 pinned .to(device, non_blocking): %.2f ms host" % ((time.perf_counter() - t) / 20 * 1e3))
torch.cuda.synchronize()
