"""Host-side training surface around the HIP step engine: what `train_student` / `train_traditional` / `validate` /
`save_checkpoint` / `resume_from_checkpoint` do around the per-step path upstream (src/optimization/train_methods.py:770-1254,
src/optimization/traditional.py:30-240), with the reference's cfg keys, checkpoint format and error behaviour.

  * optimizer   cfg `optimizer` in {SGD, Adam, AdamW}; anything else raises Exception("Unsupported optimizer ...") (:808-836).  The
                update itself is one flat HIP pass (csrc/optim.hip); its state is exported / imported in torch.optim's state_dict
                layout, parameters numbered in `named_parameters()` order, so a checkpoint written here resumes upstream and vice versa.
  * scheduler   cfg `scheduler` in {StepLR(step_size, gamma), ReduceLROnPlateau(patience=3), CosineAnnealingWarmRestarts(T_0=10)}
                else Exception("Unsupported scheduler ...") (:860-878).  torch's own scheduler objects run on a one-parameter host
                optimizer whose lr is mirrored into the engine's device-side lr, so `state_dict()` IS torch's.
  * validate    eval-mode student, per-batch sums weighted by the batch size, divided by len(val_set) (:1083-1185).
  * scalars     tensorboardX-style `all_logs.<rank>.json` with the reference's tags (traditional.py:192-236, train_methods.py:1175-1180).
"""
from __future__ import annotations

import json
import logging
import os
import shutil
import time
from typing import Dict, List, Optional

import torch

from .layout import param_rows

logger = logging.getLogger("train")

SUPPORTED_METHODS = ("traditional_nms", "traditional_nms_augmented", "traditional_nms_kdlist", "traditional_nms_kdlist_augmented")


def kdlist_augment_now(epoch: int) -> bool:
    """train_traditional's draw for `traditional_nms_kdlist_augmented` (src/optimization/traditional.py:121-124):
    random.random() > max(0.5, 0.5 + 0.5 * (1 - epoch / 50)) - never in epoch 0, half of the iterations from epoch 50 on."""
    import random
    return random.random() > max(0.5, 0.5 + 0.5 * (1 - epoch / 50))


# ------------------------------------------------------------------------------------------------ optimizer
def optimizer_settings(cfg) -> Dict:
    """cfg -> StepConfig keyword arguments; raises like the reference for an unknown optimizer."""
    name = cfg.get("optimizer", "Adam")
    if name == "SGD":
        return {"optimizer": "SGD", "momentum": cfg.getfloat("momentum"), "weight_decay": cfg.getfloat("weight_decay")}
    if name == "Adam":
        return {"optimizer": "Adam"}
    if name == "AdamW":
        return {"optimizer": "AdamW", "weight_decay": 1e-2}      # torch.optim.AdamW's default; the reference passes none
    raise Exception(f"Unsupported optimizer {name}")


def _is_head(key: str) -> bool:
    return key.startswith(("regressor", "classifier"))


def optimizer_state_dict(eng) -> Dict:
    """The engine's optimizer state as `torch.optim.{Adam,AdamW,SGD}.state_dict()` would hold it for the reference's student:
    state[i] for the i-th entry of named_parameters().  Head parameters that never received a gradient have no entry (torch
    skips `p.grad is None`), exactly like upstream before the first batch with pseudo-labels."""
    ps, cfg = eng.student.ps, eng.cfg
    rows = param_rows(eng.student.spec)
    m = ps.export_flat(eng.exp_avg)
    v = ps.export_flat(eng.exp_avg_sq) if eng.opt_mode != 2 else None
    step_main, step_head = int(eng.adam_main[0].item()), int(eng.adam_head[0].item())
    state = {}
    for i, (key, shape, _) in enumerate(rows):
        step = step_head if _is_head(key) else step_main
        if step == 0:
            continue
        if eng.opt_mode == 2:
            if cfg.momentum != 0:
                state[i] = {"momentum_buffer": m[key].reshape(shape).clone()}
        else:
            state[i] = {"step": step, "exp_avg": m[key].reshape(shape).clone(), "exp_avg_sq": v[key].reshape(shape).clone()}
    lr = eng.lr
    if eng.opt_mode == 2:
        group = {"lr": lr, "momentum": cfg.momentum, "dampening": 0, "weight_decay": cfg.weight_decay, "nesterov": False}
    else:
        group = {"lr": lr, "betas": (cfg.b1, cfg.b2), "eps": cfg.eps, "weight_decay": cfg.weight_decay if eng.opt_mode == 1 else 0,
                 "amsgrad": False}
    group["params"] = list(range(len(rows)))
    # SGD keeps no step count in torch; ours rides along so that "first step" (buf = grad) survives a resume
    return {"state": state, "param_groups": [group], "mmd_steps": {"main": step_main, "head": step_head}}


def load_optimizer_state_dict(eng, d: Dict) -> None:
    """Inverse of optimizer_state_dict; also accepts a checkpoint written by upstream torch (no `mmd_steps`) and round 1's private
    layout ({exp_avg, exp_avg_sq, step_main, step_head, head_active, lr})."""
    ps = eng.student.ps
    if "param_groups" not in d:          # round-1 layout
        ps.import_flat(eng.exp_avg, d["exp_avg"]); ps.import_flat(eng.exp_avg_sq, d["exp_avg_sq"])
        eng.adam_main[0] = d["step_main"]; eng.adam_head[0] = d["step_head"]
        eng.head_active.fill_(d["head_active"]); eng.set_lr(d["lr"])
        return
    rows = param_rows(eng.student.spec)
    groups = d["param_groups"]
    ids = [i for g in groups for i in g["params"]]
    if len(ids) != len(rows):
        raise ValueError(f"optimizer state has {len(ids)} parameters, the student has {len(rows)}")
    st = d["state"]
    m = {k: torch.zeros(s) for k, s, _ in rows}
    v = {k: torch.zeros(s) for k, s, _ in rows}
    steps = {"main": set(), "head": set()}
    for pos, (key, shape, _) in enumerate(rows):
        e = st.get(ids[pos], st.get(str(ids[pos])))
        if not e:
            continue
        if "momentum_buffer" in e:
            if e["momentum_buffer"] is not None:
                m[key] = e["momentum_buffer"].detach().float().cpu().reshape(shape)
        else:
            m[key] = e["exp_avg"].detach().float().cpu().reshape(shape)
            v[key] = e["exp_avg_sq"].detach().float().cpu().reshape(shape)
            steps["head" if _is_head(key) else "main"].add(int(float(e["step"])))
    ps.import_flat(eng.exp_avg, m); ps.import_flat(eng.exp_avg_sq, v)
    if "mmd_steps" in d:
        sm, sh = d["mmd_steps"]["main"], d["mmd_steps"]["head"]
    else:
        if len(steps["main"]) > 1 or len(steps["head"]) > 1:
            raise ValueError("per-parameter step counts differ inside the backbone / head groups: not a state this engine can hold")
        sm = steps["main"].pop() if steps["main"] else (1 if st else 0)
        sh = steps["head"].pop() if steps["head"] else 0
        if eng.opt_mode == 2:      # torch's SGD state has no step: any momentum buffer means "not the first step"
            sm = 1 if st else 0
            sh = 1 if any(_is_head(rows[p][0]) and st.get(ids[p], st.get(str(ids[p]))) for p in range(len(rows))) else 0
    eng.adam_main[0] = float(sm); eng.adam_head[0] = float(sh)
    eng.head_active.fill_(1 if sh > 0 else 0)
    eng.set_lr(float(groups[0]["lr"]))


# ------------------------------------------------------------------------------------------------ scheduler
class LrSchedule:
    """torch's scheduler classes on a one-parameter host optimizer; the engine's device-side lr follows it."""

    def __init__(self, eng, cfg):
        name = cfg.get("scheduler", "ReduceLROnPlateau")
        self.name, self.eng = name, eng
        self._p = torch.nn.Parameter(torch.zeros(1))
        self.opt = torch.optim.SGD([self._p], lr=eng.lr if eng is not None else cfg.getfloat("lr"))
        sch = torch.optim.lr_scheduler
        if name == "StepLR":
            self.sched = sch.StepLR(self.opt, step_size=cfg.getint("step_size"), gamma=cfg.getfloat("gamma"))
        elif name == "ReduceLROnPlateau":
            self.sched = sch.ReduceLROnPlateau(self.opt, patience=3)
        elif name == "CosineAnnealingWarmRestarts":
            self.sched = sch.CosineAnnealingWarmRestarts(self.opt, T_0=10)
        else:
            raise Exception(f"Unsupported scheduler {name}")

    @property
    def lr(self) -> float:
        return float(self.opt.param_groups[0]["lr"])

    def step(self, loss: float) -> None:
        """End of an epoch, as upstream (:1001-1005): StepLR steps, ReduceLROnPlateau steps on the epoch's training loss,
        CosineAnnealingWarmRestarts is constructed but never stepped."""
        if self.name == "StepLR":
            self.opt.step(); self.sched.step()
        elif self.name == "ReduceLROnPlateau":
            self.sched.step(loss)
        if self.eng is not None:
            self.eng.set_lr(self.lr)

    def state_dict(self) -> Dict:
        return self.sched.state_dict()

    def load_state_dict(self, d: Dict) -> None:
        if "num_bad_epochs" in d and "best" in d and "mode" not in d:      # round-1 private layout
            self.sched.best, self.sched.num_bad_epochs = d["best"], d["num_bad_epochs"]
            return
        self.sched.load_state_dict(d)

    def sync_lr(self, lr: float) -> None:
        """after a resume: the optimizer's lr (restored from its own state_dict) is the truth, as upstream"""
        for g in self.opt.param_groups:
            g["lr"] = lr
        if self.eng is not None:
            self.eng.set_lr(lr)


# ------------------------------------------------------------------------------------------------ checkpoints
def save_checkpoint(state: Dict, is_best: bool, cfg) -> None:
    """src/optimization/train_methods.py:1239-1254"""
    filename = f"{cfg['exp_name']}/checkpoint.{cfg['rank']}.pth.tar"
    torch.save(state, filename)
    if is_best:
        shutil.copyfile(filename, f"{cfg['exp_name']}/best.{cfg['rank']}.pth.tar")


def checkpoint_state(eng, sched: LrSchedule, epoch: int, best_loss: float, best_epoch: int) -> Dict:
    # ("drop_draws": the device-side draw counter of the drop-connect masks - an extra key upstream ignores - so a resumed run goes on
    # with fresh masks instead of replaying the first epoch's)
    return {"epoch": epoch + 1, "state_dict": eng.student.ps.export_state(), "best_loss": best_loss, "best_epoch": best_epoch,
            "optimizer": optimizer_state_dict(eng), "scheduler": sched.state_dict(), "drop_draws": int(eng.drop_state[0].item())}


def resume_from_checkpoint(cfg, eng, sched: LrSchedule):
    """src/optimization/train_methods.py:1188-1236 -> (start_epoch, best_loss, best_epoch)"""
    start_epoch, best_loss, best_epoch = 0, 1e10, 0
    path = f"{cfg['exp_name']}/checkpoint.{cfg['rank']}.pth.tar"
    if cfg.getboolean("resume", False) and os.path.exists(path):
        c = torch.load(path, map_location="cpu", weights_only=False)
        start_epoch, best_loss, best_epoch = c["epoch"], c["best_loss"], c["best_epoch"]
        eng.student.load_state(strip_module(c["state_dict"]))
        load_optimizer_state_dict(eng, c["optimizer"])
        sched.load_state_dict(c["scheduler"])
        sched.sync_lr(eng.lr)
        if "drop_draws" in c:
            eng.drop_state[0] = int(c["drop_draws"])
        logger.info(f"Starting from epoch={start_epoch}")
        logger.info(f"Load {path}")
    return start_epoch, best_loss, best_epoch


def strip_module(sd: Dict) -> Dict:
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


# ------------------------------------------------------------------------------------------------ scalars
class ScalarLog:
    """`SummaryWriter.add_scalar` + `export_scalars_to_json` of tensorboardX as the reference uses them: a dict
    {"<logdir>/<tag>": [[wall_time, step, value], ...]} written to <exp>/all_logs.<rank>.json."""

    def __init__(self, logdir: str):
        self.logdir, self.data = logdir, {}

    def add_scalar(self, tag: str, value, step: int) -> None:
        self.data.setdefault(f"{self.logdir}/{tag}", []).append([time.time(), int(step), float(value)])

    def export_scalars_to_json(self, path: str) -> None:
        with open(path, "w") as f:
            json.dump(self.data, f)

    def close(self) -> None:
        pass


# ------------------------------------------------------------------------------------------------ validation
def allreduce_mean(x: float, world: int, device) -> float:
    """Every rank must take the same scheduler / best / early-stop decision (a rank that stops alone leaves the others blocked in
    the gradient all-reduce): the scalar the decision hangs on is averaged over the ranks first."""
    if world <= 1:
        return float(x)
    import torch.distributed as dist
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return float(t.item()) / world


def validate(eng, val_set, cfg, epoch: int, writer: Optional[ScalarLog], to_batch, collate_fn, world: int = 1) -> float:
    """validate() of src/optimization/train_methods.py:1083-1185: student in eval mode, batch_size = min(cfg batch_size, len),
    drop_last, per-batch losses (reg + cls + w_kd * sum(kd)) times the batch's sample count, divided by len(val_set)."""
    from torch.utils.data import DataLoader
    n = len(val_set)
    loader = DataLoader(val_set, batch_size=min(cfg.getint("batch_size"), n), shuffle=False, drop_last=True,
                        collate_fn=collate_fn, num_workers=cfg.getint("num_workers", 0))
    w_main, w_kd = cfg.getfloat("w_main", 1.0), cfg.getfloat("w_kd", 0.005)
    tot = reg_t = cls_t = kd_t = 0.0
    for item in loader:
        batch = to_batch(item)
        ns = batch["audio"].shape[0]
        reg, cls, kd = eng.eval_losses(batch)
        tot += (w_main * (reg + cls) + w_kd * kd) * ns
        reg_t += reg * ns; cls_t += cls * ns; kd_t += kd * w_kd * ns
    val_loss, val_reg, val_cls, val_kd = tot / n, reg_t / n, cls_t / n, kd_t / n
    val_loss = allreduce_mean(val_loss, world, eng.device)
    logger.warning("=" * 15 + "VAL" + "=" * 15)
    logger.warning(f"Epoch: {epoch + 1}/{cfg.getint('num_epoches', 1)}")
    logger.warning(f"Loss:{val_loss}")
    logger.warning(f"Regression:{val_reg}")
    logger.warning(f"Cls:{val_cls}")
    logger.warning("KLDiv:0")
    logger.warning(f"KD:{val_kd}")
    logger.warning("=" * 34)
    if writer:
        writer.add_scalar("Test/Total_loss", val_loss, epoch)
        writer.add_scalar("Test/Regression_loss", val_reg, epoch)
        writer.add_scalar("Test/Class_loss", val_cls, epoch)
        writer.add_scalar("Test/KLDiv", 0.0, epoch)
        writer.add_scalar("Test/KD", val_kd, epoch)
    return val_loss
