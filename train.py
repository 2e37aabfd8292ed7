#!/usr/bin/env python3
"""Training entry point with the reference's CLI and config surface (train.py:223-316 upstream):

    python train.py --config_file configs/mm-distillnet.cfg [--overwrite '{"k": v}'] [--rank r --local_rank r --nodes n]

One process per GPU.  With `engine=DistributedDataParallel` (or WORLD_SIZE>1 from torchrun) the ranks
rendezvous at MASTER_ADDR/MASTER_PORT (default 127.0.0.1:23457 as upstream) over RCCL ("nccl") and exchange
student gradients only.  The step itself is mm_distillnet_amd.step.DistillEngine (HIP kernels, graph-captured).
Checkpoints keep the upstream format: <exp>/checkpoint.<rank>.pth.tar / best.<rank>.pth.tar with keys
epoch, state_dict, best_loss, best_epoch, optimizer, scheduler (src/optimization/train_methods.py:1049-1064,1239-1254); `optimizer`
is a torch.optim state_dict over named_parameters() order and `scheduler` torch's scheduler state_dict (mm_distillnet_amd/trainer.py),
so upstream can resume from these files and this script from upstream's.  Epoch logic as upstream (:966-1064): scheduler step on the
epoch's last training loss, validate() every val_interval epochs, best / early stopping on the validation loss.
"""
import argparse
import configparser
import json
import logging
import os
import shutil
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

from mm_distillnet_amd.arch import make_spec  # noqa: E402
from mm_distillnet_amd.data import (SyntheticMultimodalDetection, RawSyntheticMultimodalDetection, DeviceInputPipeline, TensorInputPipeline,  # noqa: E402
                                    CachedBatches, collate, collate_raw, valid_classes_dict)
from mm_distillnet_amd import _lib  # noqa: E402
from mm_distillnet_amd import trainer as TR  # noqa: E402
from mm_distillnet_amd.model import filter_state_dict  # noqa: E402
from mm_distillnet_amd.step import DistillEngine, StepConfig  # noqa: E402
from mm_distillnet_amd.synth import synth_state  # noqa: E402

logger = logging.getLogger("train")
_MOD_CH = {"rgb": 3, "depth": 3, "thermal": 1}
_MOD_PATH = {"rgb": "trained_models/yet-another-efficientdet-d2-rgb.pth", "depth": "trained_models/yet-another-efficientdet-d2-depth.pth",
             "thermal": "trained_models/yet-another-efficientdet-d2-thermal.pth"}


def parse_config(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config_file", required=True)
    ap.add_argument("--overwrite", type=str, default=None, help="JSON dict of cfg overrides")
    ap.add_argument("--rank", type=int, default=int(os.environ.get("RANK", 0)))
    ap.add_argument("--local_rank", type=int, default=int(os.environ.get("LOCAL_RANK", 0)))
    ap.add_argument("--nodes", type=int, default=1)
    ap.add_argument("--max_steps", type=int, default=-1, help="stop after this many optimizer steps (smoke runs)")
    args = ap.parse_args(argv)
    cp = configparser.ConfigParser()
    if not cp.read(args.config_file):
        raise Exception(f"Cannot read config file {args.config_file}")
    cfg = cp["DEFAULT"]
    if args.overwrite:
        for k, v in json.loads(args.overwrite).items():
            cfg[k] = str(v)
    cfg["rank"] = str(args.rank)
    cfg["local_rank"] = str(args.local_rank)
    cfg["nodes"] = str(args.nodes)
    return cfg, args


LAST_RUN_STEPS = 0        # optimizer steps of the most recent main() (read by the tests)


def step_config(cfg) -> StepConfig:
    method = cfg.get("train_method", "traditional_nms_augmented")
    if method not in TR.SUPPORTED_METHODS:
        # the adversarial / generator / BOHB methods are outside the hot path.
        # Upstream: raise Exception(f"Unsupported train method ...") (src/optimization/train_methods.py:1000)
        raise Exception(f"Unsupported train method {method} provided")
    # valid_labels -> VOC ids of the classes a teacher prediction must have (src/datasets/BaseDataset.py:141-165, utils.py:285-323)
    vl = cfg.get("valid_labels", "car")
    vcd = valid_classes_dict(tuple(v.strip() for v in vl.split(",")) if vl else None)
    return StepConfig(image_size=cfg.getint("image_size"), conf_threshold=cfg.getfloat("conf_threshold", 0.3),
                      nms_threshold=cfg.getfloat("nms_threshold", 0.5), T=float(cfg.get("T", 9)), p=float(cfg.get("p", 2)),
                      w_main=cfg.getfloat("w_main", 1.0), w_kd=cfg.getfloat("w_kd", 0.005), lr=cfg.getfloat("lr", 1e-4),
                      b1=cfg.getfloat("b1", 0.9), b2=cfg.getfloat("b2", 0.999), grad_clip=cfg.getfloat("grad_clip", -1),
                      kd_mode="list" if "kdlist" in method else "pairwise",
                      valid_prediction_ids=tuple(sorted(vcd["predictions_i2txt"].keys())),
                      # src/optimization/traditional.py:136: augment = config.getboolean('audio_augmentation_merge'); only
                      # ModelWithNMSLossAugmented acts on it (key absent from the shipped cfg -> off)
                      augment=bool(cfg.getboolean("audio_augmentation_merge", False)) and method == "traditional_nms_augmented",
                      # extension key (absent from the reference's cfg files -> fp32): "bf16" = 1x1 convs on the bf16 MFMA
                      precision=cfg.get("precision", "fp32"), seed=cfg.getint("seed", 24),
                      **TR.optimizer_settings(cfg))


def load_states(cfg, coef=2):
    """Reference weight files when present (key-remapped like filter_model_dict), deterministic synthetic weights otherwise."""
    mods = [m for m in ("rgb", "depth", "thermal") if m == "rgb" and cfg.getboolean("use_rgb", True) or
            m != "rgb" and cfg.getboolean(f"use_{m}", True)]
    tspecs = {m: make_spec(coef, _MOD_CH[m]) for m in mods}
    tstates = {}
    for i, m in enumerate(mods):
        # synthetic_cls_bias (float, optional): classifier-header bias of the synthetic stand-in teachers, so that they emit
        # pseudo-labels (the default -4 emits none and only the KD term trains)
        st = synth_state(tspecs[m], seed=101 + i, cls_bias=cfg.getfloat("synthetic_cls_bias", -4.0))
        if os.path.exists(_MOD_PATH[m]):
            st.update(filter_state_dict({k: v.shape for k, v in st.items()}, torch.load(_MOD_PATH[m], map_location="cpu")))
        else:
            logger.warning("teacher weights %s not found: using deterministic synthetic weights", _MOD_PATH[m])
        tstates[m] = st
    sspec = make_spec(coef, 8)
    sstate = synth_state(sspec, seed=7)
    for pth in ["trained_models/yet-another-efficientdet-d2.pth", "trained_models/yet-another-efficientdet-d2-embedding.pth"]:
        if os.path.exists(pth):
            sstate.update(filter_state_dict({k: v.shape for k, v in sstate.items()}, torch.load(pth, map_location="cpu")))
    return sspec, sstate, tspecs, tstates


class AsyncScalars:
    """The loop's loss read-back without a host sync: [reg, cls, kd sum, lr, overflow flag] go to pinned memory behind the step that produced
    them (a non-blocking D2H copy + an event on the compute stream) and are logged when the NEXT read-back is posted, ten steps later (or at
    the end of the epoch) - by then the copy has long finished, so the host never waits for the GPU to drain.  (The reference reads
    `loss.item()` every iteration, src/optimization/traditional.py:171-190; three .item() calls + check_overflow were four syncs per log.)"""

    def __init__(self):
        self.buf = torch.empty(6, dtype=torch.float32).pin_memory()
        self.ev, self.meta = None, None

    def post(self, eng, out, meta):
        vec = torch.stack([out["reg"].reshape(-1)[0], out["cls"].reshape(-1)[0], out["kd"].sum(), eng.hyper[0].float(),
                           eng.overflow.reshape(-1)[0].float(), out["nbox"].float().mean()])
        self.buf.copy_(vec, non_blocking=True)
        self.ev = torch.cuda.Event()
        self.ev.record()
        self.meta = meta

    def take(self):
        """-> ((reg, cls, kd, lr, overflow), meta) of the pending post, or None."""
        if self.ev is None:
            return None
        self.ev.synchronize()
        vals, meta = self.buf.tolist(), self.meta
        self.ev, self.meta = None, None
        return vals, meta


def main(argv=None):
    cfg, args = parse_config(argv)
    if _lib.work_skipping_switches():
        raise Exception("work-skipping dev switches are set (%s): train.py refuses to start - every gradient would be wrong"
                        % ", ".join(_lib.work_skipping_switches()))
    rank, local = args.rank, args.local_rank
    ngpu = cfg.getint("ngpu", 1)
    world = int(os.environ.get("WORLD_SIZE", ngpu * args.nodes if cfg.get("engine") == "DistributedDataParallel" else 1))
    from mm_distillnet_amd.hostinfo import cpu_share
    # host-side torch ops on this rank's part of the CPU share (cgroup quota), not one thread per visible core
    torch.set_num_threads(max(1, cpu_share() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", ngpu if world > 1 else 1)))))
    os.makedirs(cfg["exp_name"], exist_ok=True)
    logging.basicConfig(level=logging.INFO, handlers=[logging.StreamHandler(),
                                                      logging.FileHandler(f"{cfg['exp_name']}/{os.path.basename(cfg['exp_name'].rstrip('/'))}.{rank}.log")])
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "23457")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.manual_seed(cfg.getint("seed", 24))
    scfg = step_config(cfg)                      # raises on an unsupported train_method / optimizer, like upstream
    sspec, sstate, tspecs, tstates = load_states(cfg, int(cfg.get("compound_coef", 2)))
    if cfg.getint("synthetic_teacher_candidates", 0) > 0:
        # extension key (synthetic stand-in teachers only): shift each teacher's classifier bias so that about this many candidates per
        # image pass the confidence threshold on the first training samples - bench.py's workload (random-weight teachers otherwise emit
        # nothing or thousands of boxes per image, and the pseudo-label / loss phase of the step is then not the headline workload's)
        from mm_distillnet_amd.synth import tune_teacher_bias
        probe = SyntheticMultimodalDetection(cfg, "train")
        nb = min(cfg.getint("batch_size"), len(probe))
        first = collate([probe[i] for i in range(nb)])
        xin = {"rgb": first[0], "thermal": first[1], "depth": first[2]}
        for m in tspecs:
            if not os.path.exists(_MOD_PATH[m]):
                tune_teacher_bias(tspecs[m], tstates[m], xin[m], dev, cfg.getint("synthetic_teacher_candidates"))
    eng = DistillEngine(sspec, tspecs, dev, scfg, world_size=world)
    eng.load(sstate, tstates)
    sched = TR.LrSchedule(eng, cfg)              # raises on an unsupported scheduler, like upstream
    if world > 1:
        import torch.distributed as dist
        dist.broadcast(eng.student.ps.flat, 0)
        eng.student.refresh()
        if os.environ.get("MMD_COMM") == "rccl":
            eng.init_comm(rank)      # exchange through the C ABI's RCCL communicator (csrc/comm.hip) instead of torch.distributed
    # input_pipeline = raw: samples arrive as decoded frames (uint8/uint16/float mel stacks) and Normalizer/Resizer/transposes
    # run on the GPU on a copy stream (mm_distillnet_amd.data.DeviceInputPipeline); default: ready-made tensors
    raw = cfg.get("input_pipeline", "tensor") == "raw"
    Set = RawSyntheticMultimodalDetection if raw else SyntheticMultimodalDetection
    train_set, val_set = Set(cfg, "train"), Set(cfg, "val")
    pipe = DeviceInputPipeline(cfg.getint("image_size"), dev) if raw else TensorInputPipeline(dev)
    vpipe = DeviceInputPipeline(cfg.getint("image_size"), dev) if raw else None
    collate_fn = collate_raw if raw else collate
    sampler = torch.utils.data.distributed.DistributedSampler(train_set, num_replicas=world, rank=rank) if world > 1 else None
    loader = torch.utils.data.DataLoader(train_set, batch_size=cfg.getint("batch_size"), shuffle=sampler is None,
                                         drop_last=True, collate_fn=collate_fn,
                                         num_workers=cfg.getint("num_workers", 0), sampler=sampler, pin_memory=True)
    if cfg.getint("synthetic_cache", 0) > 0:
        # extension key: the first N batches generated once, kept pinned and cycled - the loop is what is measured, not the generator
        loader = CachedBatches(loader, cfg.getint("synthetic_cache"))

    def to_batch(item):
        if raw:
            return vpipe.submit(item).wait()
        rgb, thermal, depth, audio, label, ids = item
        return {"rgb": rgb.to(dev, non_blocking=True), "thermal": thermal.to(dev, non_blocking=True),
                "depth": depth.to(dev, non_blocking=True), "audio": audio.to(dev, non_blocking=True)}

    start_epoch, best_loss, best_epoch = TR.resume_from_checkpoint(cfg, eng, sched)
    writer = TR.ScalarLog(cfg["exp_name"])
    w_main, w_kd = cfg.getfloat("w_main", 1.0), cfg.getfloat("w_kd", 0.005)
    n_epochs = cfg.getint("num_epoches", 1)
    kdlist_aug = cfg.get("train_method", "") == "traditional_nms_kdlist_augmented"
    kdlist_pass = kdlist_aug and cfg.getboolean("audio_augmentation_merge", False)
    if kdlist_aug and raw:
        raise Exception("traditional_nms_kdlist_augmented needs the dataset's yield_batch: not available with input_pipeline = raw")
    no_validation = cfg.getboolean("no_validation", False)
    steps, captured, loss, val_loss = 0, False, float("nan"), float("nan")
    stop = False
    for epoch in range(start_epoch, n_epochs):
        if sampler is not None:
            sampler.set_epoch(epoch)
        t0, n_img, out = time.time(), 0, None
        t_50, n_50 = None, 0
        # MMD_TRAIN_TIMING=1: host seconds per segment of the loop (loader / input wait / capture+replay / submit / logging), printed per epoch
        seg = {"loader": 0.0, "input": 0.0, "step": 0.0, "submit": 0.0, "log": 0.0} if os.environ.get("MMD_TRAIN_TIMING") else None
        tick = time.perf_counter
        it = iter(loader)
        # one batch of look-ahead on either input path: batch n+1 is copied (and, raw path, transformed) on the pipeline's copy stream while
        # the step of batch n runs
        nxt = next(it, None)
        staged = pipe.submit(nxt) if nxt is not None else None
        num_iter = len(loader)
        i_iter = 0
        scal = AsyncScalars()
        inflight = [None, None]          # backpressure: the host stays at most two steps ahead of the GPU

        def emit(taken):
            nonlocal loss
            (reg, cls, kd, lr, ovf, nbox), (ep, it_no, gstep) = taken
            if ovf:
                eng.check_overflow()         # raises with the configured capacities
            loss = w_main * (reg + cls) + w_kd * kd
            logger.info("Epoch: %d/%d Iteration: %d/%d Lr: %g Loss:%.5f Regression:%.5f Cls:%.5f KLDiv:0 KD:%.5f PseudoLabels/img:%.1f", ep + 1,
                        n_epochs, it_no + 1, num_iter, lr, loss, reg, cls, kd, nbox)
            writer.add_scalar("Train/Total_loss", loss, gstep); writer.add_scalar("Train_/Regression_loss", reg, gstep)
            writer.add_scalar("Train/Class_loss", cls, gstep); writer.add_scalar("Train/KLDiv", 0.0, gstep)
            writer.add_scalar("Train/KD", kd, gstep)

        while nxt is not None:
            item = nxt
            ts = tick()
            batch = staged.wait()
            if seg is not None:
                seg["input"] += tick() - ts
            if kdlist_aug and TR.kdlist_augment_now(epoch):
                # traditional_nms_kdlist_augmented (traditional.py:121-124, train_methods.py:50-162): the batch's audio is replaced by
                # its mix with other recordings' audio (`label, audio = train_set.yield_batch(...)`).  Whether those recordings' RGB frames
                # then go through the RGB teacher as a 4th list entry is NOT decided by this draw: upstream hands the model
                # augment=cfg audio_augmentation_merge (traditional.py:136), so with the flag off only the mixed audio is used.
                # (Flag on and no draw: upstream would feed the dataset's `label` to the RGB teacher - not reproduced, see INTEGRATION.md.)
                aug_rgb, mixed = train_set.yield_batch(batch["audio"].shape[0], item[5])
                batch = dict(batch, audio=mixed.to(dev, non_blocking=True))
                if kdlist_pass:
                    batch["aug_rgb"] = aug_rgb.to(dev, non_blocking=True)
            ts = tick()
            nxt = next(it, None)
            t1 = tick()
            if inflight[steps & 1] is not None:
                inflight[steps & 1].synchronize()        # step n-2 has finished (normally long ago: no wait)
            if not captured:
                eng.capture(batch); captured = True
            out = eng.replay(batch)
            ev = torch.cuda.Event(); ev.record(); inflight[steps & 1] = ev
            t2 = tick()
            if nxt is not None:
                staged = pipe.submit(nxt)        # after replay(): the static inputs were copied out of `batch` on the compute stream
            t3 = tick()
            n_img += batch["audio"].shape[0]; steps += 1
            if steps % 10 == 0 or steps == 1:
                prev = scal.take()               # posted ten steps ago: its copy has finished, no wait
                if prev is not None:
                    emit(prev)
                scal.post(eng, out, (epoch, i_iter, epoch * num_iter + i_iter))
            if seg is not None:
                seg["loader"] += t1 - ts; seg["step"] += t2 - t1; seg["submit"] += t3 - t2; seg["log"] += tick() - t3
            if i_iter + 1 == 50:
                torch.cuda.synchronize(); t_50, n_50 = time.time(), n_img      # steady-state rate: the first 50 steps (capture, worker start-up) excluded
            i_iter += 1
            if 0 < args.max_steps <= steps:
                stop = True
                break
        last = scal.take()
        if last is not None:
            emit(last)
        torch.cuda.synchronize()
        t_end = time.time()
        eng.check_overflow()
        if out is None:
            logger.warning("epoch %d: the loader yielded no batch (dataset smaller than batch_size with drop_last)", epoch + 1)
            break
        # train_traditional returns the LAST iteration's loss (traditional.py:238); averaged over the ranks so that every rank
        # feeds its scheduler the same number
        loss = w_main * (out["reg"].item() + out["cls"].item()) + w_kd * out["kd"].sum().item()
        loss = TR.allreduce_mean(loss, world, dev)
        logger.info("epoch %d: %.1f images/sec on this rank, last loss %.5f", epoch + 1, n_img / (time.time() - t0), loss)
        if seg is not None:
            logger.info("epoch %d host seconds: %s (wall %.2f, %d steps)", epoch + 1, {k: round(v, 3) for k, v in seg.items()}, time.time() - t0, i_iter)
        if t_50 is not None and n_img > n_50:
            logger.info("epoch %d steady state (iterations 51..%d, device synchronised at both ends): %.1f images/sec = %.2f ms/step", epoch + 1,
                        i_iter, (n_img - n_50) / (t_end - t_50), (t_end - t_50) / ((n_img - n_50) / batch["audio"].shape[0]) * 1e3)
        sched.step(loss)
        if no_validation:
            if stop:
                break
            continue
        is_best = False
        if epoch % cfg.getint("val_interval", 1) == 0:
            val_loss = TR.validate(eng, val_set, cfg, epoch, writer, to_batch, collate_fn, world)
            is_best = val_loss < best_loss
            if is_best:
                torch.save(eng.student.ps.export_state(), f"{cfg['exp_name']}/only_parameters_student_best.{rank}")
                best_loss, best_epoch = val_loss, epoch + 1
            if epoch - best_epoch > cfg.getint("es_patience", 5) > 0:
                logger.info(f"ES Epoch{epoch}. Lowest loss is {val_loss}")
                break
        if not (cfg.getboolean("fast_run", False) and not is_best):
            TR.save_checkpoint(TR.checkpoint_state(eng, sched, epoch, best_loss, best_epoch), is_best, cfg)
        if stop:           # --max_steps reached: checked on every path out of the epoch body
            break
    global LAST_RUN_STEPS
    LAST_RUN_STEPS = steps
    writer.export_scalars_to_json(f"{cfg['exp_name']}/all_logs.{rank}.json")
    try:
        if no_validation and captured:
            val_loss = TR.validate(eng, val_set, cfg, n_epochs, writer, to_batch, collate_fn, world)
    finally:
        eng.close_comm()        # the C-ABI RCCL communicator (MMD_COMM=rccl) goes before the process group does
        if world > 1:
            import torch.distributed as dist
            dist.barrier(); dist.destroy_process_group()
    return val_loss if val_loss == val_loss else loss


if __name__ == "__main__":
    main()
