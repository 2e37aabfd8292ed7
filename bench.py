#!/usr/bin/env python3
"""Headline benchmark: distillation-step images/sec (3 frozen EfficientDet-D2 teachers + trainable audio
student, MTA + focal losses, backward, gradient all-reduce, Adam), D2 @ 512x512, per-GPU batch 8, fp32.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; weak scaling (per-GPU batch fixed); student gradients all-reduced with RCCL.
Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline` for the dominant kernel
family (hipEvents on the launch stream, see csrc/prof.hip) and `cpu_baseline` (the oracle/ port of the same
step on the host cores, bounded sample, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time


# Hardware queues (opt-in): the step's graph has six concurrent branches (main chain, three teachers, weight gradients,
# regressor head); the HIP runtime multiplexes a process' streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).
# MMD_HW_QUEUES=8 measured 21.35 ms/step (mean of 12 runs) against 21.73 with the default; left off by default because the
# runtime proved fragile away from its default (2 queues: 40.7 ms/step or a segfault inside graph replay) - profiles/r01_notes.md.
# Must be in the environment before the HIP runtime initialises, i.e. before torch is imported.
if os.environ.get("MMD_HW_QUEUES"):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ["MMD_HW_QUEUES"])
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mm_distillnet_amd import _lib  # noqa: E402
from mm_distillnet_amd.arch import make_spec  # noqa: E402
from mm_distillnet_amd.step import DistillEngine, StepConfig  # noqa: E402
from mm_distillnet_amd.synth import synth_state, synth_inputs, calibrated_state, tune_teacher_bias  # noqa: E402

FAMILIES = {0: ("pw_gemm_kernel (1x1 conv fwd / input-grad MFMA GEMM)", "mfma"),
            1: ("pw_wgrad_kernel (1x1 conv weight-grad MFMA GEMM)", "mfma"),
            2: ("dw_fwd_kernel (depthwise conv forward)", "hbm"),
            3: ("dw_bwd kernels (depthwise conv backward)", "hbm"),
            4: ("row-streaming kernels (BN backward / affine / pools)", "hbm"),
            5: ("mbx_kernel / bifpn_node_fused_kernel (frozen nets: expand + depthwise, whole BiFPN node, one kernel each)", "hbm"),
            6: ("se_* kernels (squeeze-excite FCs: forward + data gradients; launch-latency bound)", "hbm"),
            7: ("fuse_dw_bwd_kernel (BiFPN node backward: depthwise + fusion [+ 1x1 input gradient])", "hbm")}
FAM_KEY = {0: "pw_gemm", 1: "pw_wgrad", 2: "dw_fwd", 3: "dw_bwd", 4: "bn_bwd", 5: "mbx", 6: "se", 7: "node_bwd"}
PEAK = {"mfma": 157.3, "hbm": 8000.0}      # TFLOP/s fp32 MFMA, GB/s HBM3E (MI355X_MICROARCH.md)


def cpu_baseline(sstate, tstates, S, sample_b, coef=2):
    from oracle import step_ref as ST
    from mm_distillnet_amd.arch import make_spec as ms
    from mm_distillnet_amd.hostinfo import limit_torch_threads
    limit_torch_threads(64)         # this process' CPU share (affinity and cgroup quota), at most 64
    log("cpu baseline on %d threads" % torch.get_num_threads())
    batch = synth_inputs(sample_b, S, seed=77)
    spec = ms(coef, 8)
    ones = {b.idx: torch.ones(sample_b) for b in spec.blocks if b.skip}

    def one():
        st = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
              for k, v in sstate.items()}
        t0 = time.time()
        out = ST.distill_forward(st, tstates, batch, S, coef, ones)
        loss = ST.total_loss(out)
        if loss.requires_grad:
            loss.backward()
        params = {k: v for k, v in st.items() if v.requires_grad}
        grads = {k: v.grad for k, v in params.items() if v.grad is not None}
        with torch.no_grad():
            ST.adam_step(params, grads, {})
        return time.time() - t0

    t_w = one()                 # warm-up (thread pools, allocator)
    log("cpu baseline warm-up step %.1fs" % t_w)
    # BASELINE.md section 3 / SURVEY 8(d): the full per-GPU batch, median of >= 3 steps after 1 warm-up (bounded: 2 timed steps if one
    # step takes more than 40 s on this host)
    ts = sorted(one() for _ in range(3 if t_w <= 40 else 2))
    med = ts[len(ts) // 2]
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    log("cpu baseline steps %s s on %s" % ([round(t, 1) for t in ts], model))
    return {"value": round(sample_b / med, 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": model, "steps_s": [round(t, 2) for t in ts],
            "sample": f"{sample_b} images @ {S}x{S} (one full per-GPU batch): the whole step (3 teacher fwd + student fwd/bwd + losses "
                      f"+ Adam) of the oracle/ PyTorch-CPU restatement - kind 'port': NOT the reference's own code, which cannot "
                      f"travel to this box; pinned against it by tests/golden - median of {len(ts)} after 1 warm-up"}


_T0 = time.time()


def launch_ranks(args, argv=None, runner=None):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process becomes the launcher - it starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <same arguments>` as a CHILD process (one rank per GPU,
    the reference's `train.py:296-313` mp.spawn shape), lets rank 0's single JSON line through on stdout and returns the child's exit
    status.  Nothing here touches the GPU (no HIP call before the child exists, and no exec).  Returns None when this process is a rank
    itself; a rank whose WORLD_SIZE disagrees with --gpus refuses with exit status 2 instead of printing a line for the wrong N."""
    import subprocess
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != args.gpus:
            print("bench.py: --gpus %d but WORLD_SIZE=%s: start one rank per GPU (python -m torch.distributed.run --nproc-per-node %d "
                  "bench.py --gpus %d ...) or run `python bench.py --gpus %d` and let it start them" % (
                      args.gpus, world_env, args.gpus, args.gpus, args.gpus), file=sys.stderr, flush=True)
            return 2
        return None
    if args.gpus <= 1:
        return None
    import socket
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    log("launcher: starting %d ranks: %s" % (args.gpus, " ".join(cmd)))
    return (runner or subprocess.run)(cmd, env=env).returncode


def alt_leg_wanted(args, env, skipping=()) -> bool:
    """The second (v_mfma_f32) leg runs only for the plain full-record fp32 command of ONE process that no profiler has preloaded into."""
    profiled = "rocprof" in env.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCP", "ROCPROF")) for k in env)
    return bool(args.precision == "fp32" and not args.no_alt and not args.no_cpu_baseline and not skipping and not profiled
                and not env.get("MMD_MFMA_F32") and int(env.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1)


def alt_mfma_f32(args):
    """The default fp32 workload once more in a child process with MMD_MFMA_F32=1 -> {ms_per_step, value, gemm family ms / frac} or {error}."""
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup), "--batch", str(args.batch),
           "--size", str(args.size), "--coef", str(args.coef), "--precision", "fp32", "--no-cpu-baseline", "--no-alt"]
    if args.no_graph:
        cmd.append("--no-graph")
    import subprocess
    env = dict(os.environ, MMD_MFMA_F32="1")
    env.pop("MMD_PROF_DUMP", None)
    log("alt leg: the same workload with MMD_MFMA_F32=1 in a child process")
    try:
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300)
        rec = json.loads(out.stdout.decode().strip().splitlines()[-1])
        r = rec.get("roofline") or {}
        return {"what": "same command, MMD_MFMA_F32=1: every GEMM kernel on v_mfma_f32_32x32x2_f32 (child process, run before this one's GPU work)",
                "ms_per_step": rec.get("ms_per_step"), "value": rec.get("value"), "unit": rec.get("unit"),
                "gemm_family_ms_per_step": r.get("family_ms_per_step"), "gemm_family_frac": r.get("frac")}
    except Exception as e:      # the record must not depend on the second leg
        return {"error": repr(e)[:300]}


def log(msg):
    print("[bench %6.1fs] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--coef", type=int, default=2, help="EfficientDet compound coefficient (2 = the BASELINE configs 1-4; 4 with --size 768 = config 5's architecture, run in fp32)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"],
                    help="bf16 = mixed precision (BASELINE config 5): 1x1-conv GEMMs on the bf16 MFMA, fp32 accumulate; the default workload is fp32 like the "
                         "reference (the bf16_hbm storage mode was deleted in round 6: no faster for three rounds)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-alt", action="store_true",
                    help="skip the second measurement of the fp32 record (the same command in a child process with MMD_MFMA_F32=1: every GEMM on v_mfma_f32)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="images in the CPU baseline's batch (0 = the per-GPU batch)")
    ap.add_argument("--dev-timing", action="store_true",
                    help="run although work-skipping dev switches are set (MMD_DEV=1 + MMD_DEV_SKIP_CALLS / _SKIP_WG / _NO_BWD / MMD_ROWS_ABL): the record "
                         "then carries \"invalid\": true and NO value")
    args = ap.parse_args()
    # self-certifying record: a run with launches skipped is not a measurement - refuse it before anything else happens
    skipping = _lib.work_skipping_switches()
    if skipping and not args.dev_timing:
        print("bench.py: work-skipping dev switches are set (%s): the timed region would not do the work - refusing.  Unset them, or pass "
              "--dev-timing for a timing experiment (the record then says \"invalid\": true and has no value)" % ", ".join(skipping),
              file=sys.stderr, flush=True)
        sys.stdout.write(json.dumps({"invalid": True, "reason": "work-skipping dev switches set", "switches": skipping}) + "\n")
        sys.exit(3)
    rc = launch_ranks(args)
    if rc is not None:
        sys.exit(rc)
    # The fp32 record's second number: the same command with every GEMM on v_mfma_f32_32x32x2_f32 (MMD_MFMA_F32=1) instead of the split form
    # (fp32 products as six bf16 MFMAs on an exact three-way operand split, csrc/common.h).  The library reads the switch once per process, so
    # the measurement runs as a child process - started and finished BEFORE this process touches the GPU - and its numbers ride in the record
    # as `alt_mfma_f32`: whoever reads the line has both forms from one run on one box.
    alt = None
    # Only in the full-record mode (the CPU baseline leg on: the driver's plain `python bench.py`), never under a profiler: rocprofv3's
    # preloaded library initialises the GPU before this program starts, and a process that has done so must not start another program.
    if alt_leg_wanted(args, os.environ, skipping):
        alt = alt_mfma_f32(args)
    # stdout carries exactly ONE line, the JSON record: RCCL prints its version banner to stdout when a communicator comes up, so
    # file descriptor 1 points at stderr for the whole run and the record goes out through a saved duplicate at the end
    sys.stdout.flush()
    real_out = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MMD_FORCE_DEVICE"):      # dev aid: exercise the N>1 code path on a 1-GPU box (with MMD_DIST_BACKEND=gloo)
        local = int(os.environ["MMD_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    # host-side torch ops (weight generation, teacher calibration): this rank's part of the CPU share, not one thread per visible core
    from mm_distillnet_amd.hostinfo import cpu_share
    torch.set_num_threads(max(1, cpu_share() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world)))))
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MMD_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    S, B = args.size, args.batch
    mods = {"rgb": (3, 1), "depth": (3, 2), "thermal": (1, 3)}
    specs = {k: make_spec(args.coef, c) for k, (c, _) in mods.items()}
    calib = synth_inputs(4, 256, seed=1234)
    log("building states")
    tstates = {k: calibrated_state(specs[k], seed, calib[k], dev) for k, (_, seed) in mods.items()}
    sspec = make_spec(args.coef, 8)
    sstate = calibrated_state(sspec, 4, calib["audio"], dev)
    batch_cpu = synth_inputs(B, S, seed=24 + rank)
    # the frozen teachers are REPLICAS: their classifier biases are tuned on rank 0's batch (seed 24) and every teacher tensor is
    # broadcast from rank 0 below, so all ranks hold bit-identical teachers whatever their own batch is
    tune_batch = batch_cpu if rank == 0 else synth_inputs(B, S, seed=24)
    for k in tstates:
        tune_teacher_bias(specs[k], tstates[k], tune_batch[k], dev)
    del tune_batch
    log("teacher biases tuned")
    eng = DistillEngine(sspec, specs, dev, StepConfig(image_size=S, precision=args.precision), world_size=world, process_group=pg)
    if world == 1 and os.environ.get("MMD_FORCE_DP"):
        # dev aid for a 1-GPU box: a one-rank RCCL group, the split backward and the phased all-reduce calls exactly as at
        # N > 1 (measures what the split + the collectives' launches cost when there is nothing to exchange)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(dev))
        eng.force_ar = True
        eng.ar_split = eng._default_split()
    eng.load(sstate, tstates)
    if os.environ.get("MMD_COMM") == "rccl" and (world > 1 or eng.force_ar):
        eng.init_comm(rank)      # gradient exchange through the C ABI's own RCCL communicator (csrc/comm.hip) instead of torch.distributed
    # what the collective layer ITSELF reports (a scaling record can then show that RCCL saw N ranks, not just that N processes ran)
    collective = {"backend": None, "ranks": 1, "c_abi": False}
    if world > 1 or eng.force_ar:
        import torch.distributed as dist
        collective = {"backend": dist.get_backend(), "ranks": dist.get_world_size(), "c_abi": eng.comm is not None}
        if eng.comm is not None:
            import ctypes
            n_c = ctypes.c_int(0)
            rc = _lib.LIB.load().mmd_comm_count(eng.comm, ctypes.cast(ctypes.pointer(n_c), ctypes.c_void_p))
            collective["c_abi_ranks"] = n_c.value if rc == 0 else None
    if world > 1:   # identical initial student on every rank (DDP broadcasts parameters and buffers at construction) and identical teachers
        import torch.distributed as dist
        for net in [eng.student] + list(eng.teachers.values()):
            for t in (net.ps.flat, net.ps.rmean, net.ps.rvar):
                dist.broadcast(t, 0)
            net.refresh()
    batch = {k: v.to(dev) for k, v in batch_cpu.items()}

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = not args.no_graph
    if os.environ.get("MMD_SERIAL"):      # (dev: the whole step as ONE chain - teachers, weight gradients and the regressor branch in line -
        eng.concurrent_teachers = False   #  so that a kernel trace shows every launch alone, in order: tools/dev/trace_chain.py)
        os.environ["MMD_NO_WG"] = "1"; os.environ["MMD_NO_SIDE"] = "1"
    log("engine loaded")
    if use_graph:
        eng.capture(batch)
        log("graphs captured")
        run = lambda: eng.replay()
    else:
        run = lambda: eng.step(batch)
    for _ in range(args.warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    log("timed %d steps: %.2f ms/step" % (args.steps, ms))
    # Outside the timed region: the distribution of single synchronised steps over >= 100 more steps.  One build runs in a "fast" or a
    # "slow" mode (~5 % apart) depending on the process / time window (profiles/r02_notes.md); the 20-step mean above cannot tell which
    # one it landed in, the median and spread of 100 steps next to it can.
    per = []
    # (bounded to ~20 s of steps; `ms` is the max over the ranks, so every rank runs the same count and the per-sample barriers pair up)
    n_per = max(5, min(int(os.environ.get("MMD_BENCH_PERSTEP", "100")), int(20e3 / max(ms, 1e-3))))
    for _ in range(n_per):
        barrier() if world > 1 else torch.cuda.synchronize()
        t1 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        per.append((time.perf_counter() - t1) * 1e3)
    per.sort()
    per_step = {"n": len(per), "min_ms": round(per[0], 3), "p25_ms": round(per[len(per) // 4], 3), "median_ms": round(per[len(per) // 2], 3),
                "p75_ms": round(per[3 * len(per) // 4], 3), "max_ms": round(per[-1], 3),
                "note": "single steps, each followed by a device synchronise (adds the host's launch + sync latency to every sample)"}
    log("per-step ms: min %.3f  p25 %.3f  median %.3f  p75 %.3f  max %.3f" % (
        per[0], per[len(per) // 4], per[len(per) // 2], per[3 * len(per) // 4], per[-1]))
    if int(eng.overflow.item()):
        print("per-teacher candidate counts:", [c.cpu().tolist() for c in eng.out["cnt_t"]], file=sys.stderr)
    eng.check_overflow()
    value = world * B * args.steps / dt
    nbox = eng.out["nbox"].cpu().tolist()

    # ---- roofline of the dominant kernel family: one eager step bracketed with hipEvents on the launch stream
    roof = None
    cpu = None
    if rank == 0:
        dll = _lib.LIB.load()
        dlls = [dll]
        if os.environ.get("MMD_PROF_DUMP"):
            for i, d in enumerate(dlls):
                d.mmd_prof_dump_to((os.environ["MMD_PROF_DUMP"] + (".w16" if i else "")).encode())
        # one eager step on a single stream (teachers, weight-gradient and head side branches folded onto it), so that an
        # event pair brackets one kernel running alone - the same condition as the serialised rocprofv3 kernel trace.  Run twice, the
        # counters switched on for the second: the single-stream form launches a few kernel variants the captured schedule does not, and a
        # kernel's FIRST launch in a process uploads its code object inside the event pair (one 262 us "node backward" launch among four
        # of 21 us in profiles/r06_notes.md section 10)
        conc = eng.concurrent_teachers
        eng.concurrent_teachers = False
        os.environ["MMD_NO_WG"] = "1"; os.environ["MMD_NO_SIDE"] = "1"
        for leg in range(2):
            if leg == 1:
                for fam in FAMILIES:
                    for d in dlls:
                        d.mmd_prof_enable(fam, 1)
            torch.cuda.synchronize()
            eng.step_body(batch if not use_graph else eng.static, eng.static["drop_scale"] if use_graph else eng.make_drop_scale(B))
            eng.backward_tail()
            torch.cuda.synchronize()
        eng.concurrent_teachers = conc
        del os.environ["MMD_NO_WG"], os.environ["MMD_NO_SIDE"]
        import ctypes
        res = {}
        for fam in FAMILIES:
            tot = [0.0] * 4
            for d in dlls:
                buf = (ctypes.c_double * 4)()
                d.mmd_prof_collect(fam, buf)
                d.mmd_prof_enable(fam, 0)
                tot = [a + b for a, b in zip(tot, buf)]
            res[fam] = tot
        log("family times ms: %s" % {f: round(res[f][1], 3) for f in res})
        for d in dlls:
            d.mmd_prof_dump_to(None)
        fam = max(res, key=lambda f: res[f][1])
        std_shape = args.coef == 2 and S == 512 and B == 8
        n, tms, fl, by = res[fam]
        name, bound = FAMILIES[fam]
        if args.precision != "fp32" and bound == "mfma":
            bound = "hbm"        # on the bf16 MFMA (16x the fp32 rate) the 1x1 convs are load/store-bound
        if bound == "mfma":
            achieved = fl / (tms * 1e-3) / 1e12
            unit = "TFLOP/s"
        else:
            achieved = by / (tms * 1e-3) / 1e9
            unit = "GB/s"
        traffic = None
        fam_key = FAM_KEY[fam]
        try:      # HBM bytes per launch from the committed PMC passes of this same command (profiles/pmc_traffic.json)
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["families"][fam_key]
            # the committed PMC passes are of the default workload only
            traffic = round(pm["hbm_bytes_per_step"] / max(n, 1), 1) if (args.coef == 2 and S == 512 and B == 8 and fam_key != "bn_bwd") else None
        except Exception:
            traffic = None
        peak = PEAK[bound]
        if args.precision != "fp32":
            traffic = None          # the committed PMC passes are of the fp32 kernels
        roof = {"kernel": name, "bound": bound, "achieved": round(achieved, 3), "peak": peak, "unit": unit,
                "frac": round(achieved / peak, 4), "traffic": traffic,
                # provenance: `traffic` is NOT measured in this run (PMC counters need rocprofv3 around the process); it is the committed
                # builder-run counter pass of this same command, divided by THIS run's launch count
                "traffic_source": (None if traffic is None else
                                   "profiles/pmc_traffic.json (builder-run rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                                   "%s launches in the family there, %d here)" % (pm.get("launches", "?"), int(n))),
                "algorithmic_bytes_per_launch": round(by / max(n, 1), 1), "launches_per_step": int(n),
                "avg_launch_us": round(tms * 1e3 / max(n, 1), 2), "family_ms_per_step": round(tms, 3),
                "algorithmic_bytes_per_step": by, "algorithmic_flops_per_step": fl,
                "all_families_ms": {FAMILIES[f][0].split(" ")[0]: round(res[f][1], 3) for f in res}}
        # every profiled family against its own roof: ms = the family's launches of one eager single-stream step, achieved
        # = their algorithmic flops (mfma) / bytes (hbm) over that time, traffic_ratio = PMC HBM bytes / algorithmic bytes where the committed
        # counter passes (profiles/pmc_traffic.json, default workload) cover the family
        fams = []
        try:
            pmc_f = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["families"]
        except Exception:
            pmc_f = {}
        for f in sorted(res, key=lambda f_: -res[f_][1]):
            n_f, ms_f, fl_f, by_f = res[f]
            if not n_f:
                continue
            b_f = FAMILIES[f][1]
            if args.precision != "fp32" and b_f == "mfma":
                b_f = "hbm"
            ach = (fl_f / (ms_f * 1e-3) / 1e12) if b_f == "mfma" else (by_f / (ms_f * 1e-3) / 1e9)
            tr = None
            pk = pmc_f.get(FAM_KEY[f])
            if pk and std_shape and args.precision == "fp32" and FAM_KEY[f] != "bn_bwd" and by_f > 0:
                if FAM_KEY[f] in ("dw_fwd", "dw_bwd") and "dw_fwd" in pmc_f and "dw_bwd" in pmc_f:
                    # the depthwise forward and input-gradient launches share kernels (dw_fwd_kernel with flipped taps), so the counter
                    # passes - keyed by kernel name - cannot tell the two families apart: both report the ratio of their union
                    tr = round((pmc_f["dw_fwd"]["hbm_bytes_per_step"] + pmc_f["dw_bwd"]["hbm_bytes_per_step"]) / (res[2][3] + res[3][3]), 3)
                else:
                    tr = round(pk["hbm_bytes_per_step"] / by_f, 3)
            fams.append({"name": FAMILIES[f][0].split(" (")[0], "bound": b_f, "launches": int(n_f), "ms": round(ms_f, 3),
                         "achieved": round(ach, 3), "unit": "TFLOP/s" if b_f == "mfma" else "GB/s", "frac": round(ach / PEAK[b_f], 4),
                         "traffic_ratio": tr})
        roof["families"] = fams
        if args.precision == "fp32":
            native = bool(os.environ.get("MMD_MFMA_F32"))
            # how the fp32 GEMM kernels multiply (csrc/common.h): `achieved` counts the ALGORITHMIC fp32 flops either way
            roof["mfma_form"] = ("v_mfma_f32_32x32x2_f32 in every GEMM kernel (MMD_MFMA_F32=1)" if native else
                                 "split: LDS-tiled 1x1 kernels (K >= 64, N > 48) and the grouped weight gradient take each fp32 product as six "
                                 "v_mfma_f32_32x32x16_bf16 partial products of a three-way EXACT bf16 split of both operands, fp32 accumulate "
                                 "(dropped terms <= 2^-23 |ab|, 2^-27 rms - one fp32 rounding; 6 instead of 16 accumulator roundings per 16 k; error vs float64 not above v_mfma_f32's: tests/test_gpu_kernels.py::"
                                 "test_split3_precision); the other GEMM kernels use v_mfma_f32_32x32x2_f32")
            roof["peak_basis"] = ("157.3 TFLOP/s = dense v_mfma_f32 peak, the dtype's own pipe (MI355X_MICROARCH.md); the split form's own "
                                  "ceiling is 2516 / 6 = 419 TFLOP/s of fp32-equivalent flops on the bf16 pipe")
        if args.coef == 2 and S == 512:
            # the whole step against both roofs (3.41 GB and 55.1 GFLOP of conv-granularity work per image, cfg 3)
            sb, sf = 3.41e9 * B, 55.1e9 * B
            roof["whole_step"] = {"algorithmic_bytes": sb, "algorithmic_flops": sf, "ms": round(ms, 3),
                                  "hbm_GBps": round(sb / (ms * 1e-3) / 1e9, 1), "hbm_frac": round(sb / (ms * 1e-3) / 1e9 / PEAK["hbm"], 4),
                                  "mfma_TFLOPs": round(sf / (ms * 1e-3) / 1e12, 2),
                                  "mfma_frac": round(sf / (ms * 1e-3) / 1e12 / PEAK["mfma"], 4),
                                  "roofline_bound_ms": round(max(sb / (PEAK["hbm"] * 1e9), sf / (PEAK["mfma"] * 1e12)) * 1e3, 3)}
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(sstate, tstates, S, args.cpu_sample or B, args.coef)
        std = args.coef == 2 and S == 512 and B == 8 and args.precision == "fp32"
        line = {"metric": "distillation-step images/sec (3 teachers + audio student, D%d, bs=%d)" % (args.coef, B), "value": round(value, 2),
                "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32" if args.precision == "fp32" else "bf16 MFMA operands in the 1x1 convs, f32 elsewhere", "data": "synthetic",
                "config": {"workload": ("BASELINE configs[2]" if std else
                                        "BASELINE configs[4] on one GPU (D4 / 768, bf16 mixed precision: " + args.precision + ")" if (args.coef == 4 and S == 768 and args.precision != "fp32") else
                                        "non-default shape / precision (BASELINE config 5 = D4 / 768 in bf16)")
                                       + ": full 3-teacher (RGB+thermal+depth) -> audio student distillation step, EfficientDet-D%d, "
                                         "%dx%d, per-GPU batch %d, fwd+losses+bwd+all-reduce+Adam" % (args.coef, S, S, B),
                           "global_batch": world * B, "image_size": S, "parallelism": "dp%d" % world,
                           "graph": use_graph, "pseudo_label_boxes_per_image": nbox, "collective": collective,
                           # every MMD_* variable this process saw (A/B knobs select kernels / schedules; the work-skipping ones are refused above)
                           "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("MMD_")}},
                "per_step": per_step, "roofline": roof, "cpu_baseline": cpu}
        if alt is not None:
            line["alt_mfma_f32"] = alt
        if skipping:        # --dev-timing: launches were skipped - not a measurement
            line["invalid"] = True
            line["invalid_reason"] = "work-skipping dev switches set: " + ", ".join(skipping)
            line["dev_ms_per_step"] = line.pop("ms_per_step")
            line["value"] = None
        sys.stdout.flush()
        os.write(real_out, (json.dumps(line) + "\n").encode())
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
