"""Host CPU share of this process: min(scheduler affinity, cgroup CPU quota).  A GPU box hands a container a quota (e.g. 16 of 256 cores,
cgroup `cpu.max`) that neither `os.cpu_count()` nor the affinity mask shows; torch then starts one intra-op thread per visible core and
the kernel throttles them - the oracle's CPU convolutions ran 6.5x slower on 128 threads than on 16 there (tools/dev/diag_threads.py)."""
from __future__ import annotations

import math
import os


def _cgroup_quota() -> float:
    try:                                                    # cgroup v2
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:                                                    # cgroup v1
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            return q / p
    except (OSError, ValueError):
        pass
    return math.inf


def cpu_share() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    q = _cgroup_quota()
    if q != math.inf:
        n = min(n, max(1, int(q)))
    return max(1, n)


def limit_torch_threads(cap: int | None = None) -> int:
    """torch intra-op threads = the CPU share (optionally capped); returns the count set."""
    import torch
    n = cpu_share() if cap is None else max(1, min(cpu_share(), cap))
    torch.set_num_threads(n)
    return n


def free_memory_gb() -> float:
    """Host memory this process may still take: min(what the machine has available, cgroup limit - cgroup usage), in GiB."""
    try:
        import psutil
        free = float(psutil.virtual_memory().available)
    except Exception:
        free = math.inf
    for lim, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            l = open(lim).read().strip()
            if l != "max":
                free = min(free, float(l) - float(open(cur).read().strip()))
            break
        except (OSError, ValueError):
            continue
    return free / 2 ** 30
