#!/usr/bin/env python3
"""Training entry point with the reference's CLI and config surface (train.py:223-316 upstream):

    python train.py --config_file configs/mm-distillnet.cfg [--overwrite '{"k": v}'] [--rank r --local_rank r --nodes n]

One process per GPU.  With `engine=DistributedDataParallel` (or WORLD_SIZE>1 from torchrun) the ranks
rendezvous at MASTER_ADDR/MASTER_PORT (default 127.0.0.1:23457 as upstream) over RCCL ("nccl") and exchange
student gradients only.  The step itself is mm_distillnet_amd.step.DistillEngine (HIP kernels, graph-captured).
Checkpoints keep the upstream format: <exp>/checkpoint.<rank>.pth.tar / best.<rank>.pth.tar with keys
epoch, state_dict, best_loss, best_epoch, optimizer, scheduler (src/optimization/train_methods.py:1049-1064,1239-1254).
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
from mm_distillnet_amd.data import SyntheticMultimodalDetection, RawSyntheticMultimodalDetection, DeviceInputPipeline, collate  # noqa: E402
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


def step_config(cfg) -> StepConfig:
    method = cfg.get("train_method", "traditional_nms_augmented")
    return StepConfig(image_size=cfg.getint("image_size"), conf_threshold=cfg.getfloat("conf_threshold", 0.3),
                      nms_threshold=cfg.getfloat("nms_threshold", 0.5), T=float(cfg.get("T", 9)), p=float(cfg.get("p", 2)),
                      w_main=cfg.getfloat("w_main", 1.0), w_kd=cfg.getfloat("w_kd", 0.005), lr=cfg.getfloat("lr", 1e-4),
                      b1=cfg.getfloat("b1", 0.9), b2=cfg.getfloat("b2", 0.999), grad_clip=cfg.getfloat("grad_clip", -1),
                      kd_mode="list" if "kdlist" in method else "pairwise",
                      # src/optimization/traditional.py:136: augment = config.getboolean('audio_augmentation_merge'); only
                      # ModelWithNMSLossAugmented acts on it (key absent from the shipped cfg -> off)
                      augment=bool(cfg.getboolean("audio_augmentation_merge", False)) and method == "traditional_nms_augmented",
                      # extension key (absent from the reference's cfg files -> fp32): "bf16" = 1x1 convs on the bf16 MFMA
                      precision=cfg.get("precision", "fp32"))


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


class Plateau:
    """ReduceLROnPlateau(patience=3) as configured upstream (src/optimization/train_methods.py:860-878), on the engine's lr."""

    def __init__(self, eng, patience=3, factor=0.1):
        self.eng, self.patience, self.factor = eng, patience, factor
        self.best, self.bad = float("inf"), 0

    def step(self, loss):
        if loss < self.best * (1 - 1e-4):
            self.best, self.bad = loss, 0
        else:
            self.bad += 1
            if self.bad > self.patience:
                self.eng.set_lr(float(self.eng.hyper[0].item()) * self.factor)
                self.bad = 0

    def state_dict(self):
        return {"best": self.best, "num_bad_epochs": self.bad, "patience": self.patience, "factor": self.factor}

    def load_state_dict(self, d):
        self.best, self.bad = d["best"], d["num_bad_epochs"]


def optimizer_state(eng):
    ps = eng.student.ps
    return {"exp_avg": ps.export_flat(eng.exp_avg), "exp_avg_sq": ps.export_flat(eng.exp_avg_sq),
            "step_main": float(eng.adam_main[0].item()), "step_head": float(eng.adam_head[0].item()),
            "head_active": int(eng.head_active.item()), "lr": float(eng.hyper[0].item())}


def load_optimizer_state(eng, d):
    ps = eng.student.ps
    ps.import_flat(eng.exp_avg, d["exp_avg"]); ps.import_flat(eng.exp_avg_sq, d["exp_avg_sq"])
    eng.adam_main[0] = d["step_main"]; eng.adam_head[0] = d["step_head"]
    eng.head_active.fill_(d["head_active"]); eng.set_lr(d["lr"])


def main(argv=None):
    cfg, args = parse_config(argv)
    rank, local = args.rank, args.local_rank
    ngpu = cfg.getint("ngpu", 1)
    world = int(os.environ.get("WORLD_SIZE", ngpu * args.nodes if cfg.get("engine") == "DistributedDataParallel" else 1))
    os.makedirs(cfg["exp_name"], exist_ok=True)
    logging.basicConfig(level=logging.INFO, handlers=[logging.StreamHandler(),
                                                      logging.FileHandler(f"{cfg['exp_name']}/{cfg['exp_name']}.{rank}.log")])
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "23457")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.manual_seed(cfg.getint("seed", 24))
    sspec, sstate, tspecs, tstates = load_states(cfg, int(cfg.get("compound_coef", 2)))
    eng = DistillEngine(sspec, tspecs, dev, step_config(cfg), world_size=world)
    eng.load(sstate, tstates)
    if world > 1:
        import torch.distributed as dist
        dist.broadcast(eng.student.ps.flat, 0)
        eng.student.refresh()
    # input_pipeline = raw: samples arrive as decoded frames (uint8/uint16/float mel stacks) and Normalizer/Resizer/transposes
    # run on the GPU on a copy stream (mm_distillnet_amd.data.DeviceInputPipeline); default: ready-made tensors
    raw = cfg.get("input_pipeline", "tensor") == "raw"
    train_set = RawSyntheticMultimodalDetection(cfg, "train") if raw else SyntheticMultimodalDetection(cfg, "train")
    pipe = DeviceInputPipeline(cfg.getint("image_size"), dev) if raw else None
    sampler = torch.utils.data.distributed.DistributedSampler(train_set, num_replicas=world, rank=rank) if world > 1 else None
    loader = torch.utils.data.DataLoader(train_set, batch_size=cfg.getint("batch_size"), shuffle=sampler is None,
                                         drop_last=True, collate_fn=(lambda b: b) if raw else collate,
                                         num_workers=cfg.getint("num_workers", 0), sampler=sampler, pin_memory=not raw)
    sched = Plateau(eng)
    start_epoch, best_loss, best_epoch = 0, 1e10, 0
    ckpt = f"{cfg['exp_name']}/checkpoint.{rank}.pth.tar"
    if cfg.getboolean("resume", False) and os.path.exists(ckpt):
        c = torch.load(ckpt, map_location="cpu", weights_only=False)
        start_epoch, best_loss, best_epoch = c["epoch"], c["best_loss"], c["best_epoch"]
        eng.student.load_state(c["state_dict"]); load_optimizer_state(eng, c["optimizer"]); sched.load_state_dict(c["scheduler"])
        logger.info("resumed from %s at epoch %d", ckpt, start_epoch)
    steps, captured, loss = 0, False, float("nan")
    for epoch in range(start_epoch, cfg.getint("num_epoches", 1)):
        if sampler is not None:
            sampler.set_epoch(epoch)
        t0, n_img = time.time(), 0
        for item in loader:
            if raw:
                batch = pipe.submit(item).wait()
                audio = batch["audio"]
            else:
                rgb, thermal, depth, audio, label, ids = item
                batch = {"rgb": rgb.to(dev, non_blocking=True), "thermal": thermal.to(dev, non_blocking=True),
                         "depth": depth.to(dev, non_blocking=True), "audio": audio.to(dev, non_blocking=True)}
            if not captured:
                eng.capture(batch); captured = True
            out = eng.replay(batch)
            n_img += audio.shape[0]; steps += 1
            if steps % 10 == 0 or steps == 1:
                loss = cfg.getfloat("w_main", 1.0) * (out["reg"].item() + out["cls"].item()) + cfg.getfloat("w_kd", 0.005) * out["kd"].sum().item()
                logger.info("Epoch: %d/%d Iteration: %d Lr: %g Loss:%.5f Regression:%.5f Cls:%.5f KD:%.5f", epoch + 1,
                            cfg.getint("num_epoches", 1), steps, eng.hyper[0].item(), loss, out["reg"].item(), out["cls"].item(),
                            out["kd"].sum().item())
            if 0 < args.max_steps <= steps:
                break
        torch.cuda.synchronize()
        eng.check_overflow()
        loss = cfg.getfloat("w_main", 1.0) * (out["reg"].item() + out["cls"].item()) + cfg.getfloat("w_kd", 0.005) * out["kd"].sum().item()
        logger.info("epoch %d: %.1f images/sec on this rank, last loss %.5f", epoch + 1, n_img / (time.time() - t0), loss)
        sched.step(loss)
        is_best = loss < best_loss
        if is_best:
            best_loss, best_epoch = loss, epoch
        torch.save({"epoch": epoch + 1, "state_dict": eng.student.ps.export_state(), "best_loss": best_loss, "best_epoch": best_epoch,
                    "optimizer": optimizer_state(eng), "scheduler": sched.state_dict()}, ckpt)
        if is_best:
            shutil.copyfile(ckpt, f"{cfg['exp_name']}/best.{rank}.pth.tar")
        if epoch - best_epoch > cfg.getint("es_patience", 5) > 0 or 0 < args.max_steps <= steps:
            break
    if world > 1:
        import torch.distributed as dist
        dist.barrier(); dist.destroy_process_group()
    return loss


if __name__ == "__main__":
    main()
