"""CPU, world_size 2 over gloo: the gradient exchange of the data-parallel step.  Every rank reduces the
full flat gradient buffer (sum; the 1/N average is folded into Adam) and the head_active flag (max), so a rank whose
batch produced no pseudo-labels cannot desynchronise the collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q, split):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mm_distillnet_amd.arch import make_spec
    from mm_distillnet_amd.step import DistillEngine, StepConfig
    eng = DistillEngine(make_spec(2, 8), {"rgb": make_spec(2, 3)}, "cpu", StepConfig(image_size=128), world_size=world)
    g = eng.student.ps.grad
    gen = torch.Generator().manual_seed(100 + rank)
    g.copy_(torch.randn(g.numel(), generator=gen))
    r = eng.head_ranges
    if rank == 1:                      # this rank saw no boxes: head gradients are exactly zero, flag stays 0
        g[r[0]:r[1]] = 0; g[r[2]:r[3]] = 0; g[r[4]:r[5]] = 0
    else:
        eng.head_active.fill_(1)
    mine = g.clone()
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    if split:
        # the overlapped schedule: phase 0 (heads, BiFPN, late backbone blocks) is launched asynchronously after the first
        # backward segment, phase 1 (early blocks + BatchNorm affine) after the second; phase 1 also waits for phase 0
        eng.ar_split = eng._default_split()
        (p0,), tail = eng.grad_buckets()
        eng.allreduce_grads(0)
        assert eng._ar_work is not None and len(eng._ar_work) == 2
        eng.allreduce_grads(1)
        assert eng._ar_work is None
        covered = torch.zeros(g.numel(), dtype=torch.int32)
        for b, e in [p0] + tail:
            covered[b:e] += 1
        assert bool((covered == 1).all())
    else:
        assert eng.ar_split is not None         # world_size > 1 splits by default
        eng.allreduce_grads()
    ok = bool(torch.allclose(g, both[0] + both[1]))
    q.put((rank, ok, float(g.double().sum()), int(eng.head_active.item()), eng.head_ranges, eng.student.ps.n_params))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("split", [False, True])
def test_allreduce_two_ranks(split):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, split)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, ok0, s0, a0, hr, n), (_, ok1, s1, a1, _, _) = res
    assert ok0 and ok1 and s0 == s1
    assert a0 == 1 and a1 == 1
    b0, e0, b1, e1, b2, e2 = hr
    assert 0 < b0 < e0 <= b1 < e1 <= b2 < e2 <= n and all(v % 4 == 0 for v in hr)
