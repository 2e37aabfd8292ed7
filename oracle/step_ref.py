"""ORACLE (test infrastructure only): one distillation step on the CPU.

Restates `ModelWithNMSLoss(.Augmented).forward` (src/optimization/train_methods.py:310-422 /
436-517; `augment` = the cfg key audio_augmentation_merge, absent -> falsy in the shipped cfg), the loss mixing + backward of
`train_traditional` (src/optimization/traditional.py:171-190) and torch.optim.Adam as configured in
src/optimization/train_methods.py:825-833, on top of the functional oracle net.
Teachers iterate in the reference's ModuleDict insertion order rgb -> depth -> thermal
(train.py:123-135).  Pinned by tests/golden/step_*.npz (reference step module run in the build
container through tools/oracle/make_golden.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from . import effdet_ref as net
from . import losses_ref as L
from . import postproc_ref as P

TEACHER_ORDER = ("rgb", "depth", "thermal")


def distill_forward(student: Dict[str, torch.Tensor], teachers: Dict[str, Dict[str, torch.Tensor]],
                    batch: Dict[str, torch.Tensor], image_size: int, coef: int = 2,
                    drop_masks: Optional[Dict[int, torch.Tensor]] = None, conf_threshold: float = 0.3,
                    nms_threshold: float = 0.5, T: float = 9.0, p: float = 2.0, training: bool = True,
                    kd_mode: str = "pairwise", inclusive_nms: bool = False, augment: bool = False,
                    aug_rgb: Optional[torch.Tensor] = None, per_teacher_labels: Optional[list] = None):
    """-> dict(reg[1], cls[1], kd: list of Tensor[5] per teacher (pairwise) or [Tensor[5]] (list),
               labels: merged [m,5] per image, logits_s, features_s)"""
    audio = batch["audio"]
    if augment:      # merge_batch_0_1 (train_methods.py:291-308): literal torch.pow(x, 10), floor 1e-7, log10
        audio = audio.clone()
        m = torch.pow(audio[0], 10) + torch.pow(audio[1], 10)
        m[m < 1e-7] = 1e-7
        audio[1] = torch.log10(m)
    logits_s, feats_s = net.forward(student, audio, coef, training, drop_masks)
    per_teacher, kd, feats_all = [], [], []
    B = batch["audio"].shape[0]
    # ModelWithNMSKDListLossAugmented.forward(augment=True) (src/optimization/train_methods.py:73-110): one more pass, the RGB teacher on
    # `label` = RGB frames of OTHER recordings whose audio was mixed into this batch's audio; its labels and features join the lists
    order = [m for m in TEACHER_ORDER if m in teachers] + (["augmentation"] if aug_rgb is not None else [])
    for mod in order:
        with torch.no_grad():
            if mod == "augmentation":
                logits_t, feats_t = net.forward(teachers["rgb"], aug_rgb, coef, False)
            else:
                logits_t, feats_t = net.forward(teachers[mod], batch[mod], coef, False)
            feats_t = [f.detach() for f in feats_t]
            if augment:      # average_batch_0_1 (:279-289)
                feats_t = [f.clone() for f in feats_t]
                for f in feats_t:
                    f[1] = (f[0] + f[1]) / 2
            if per_teacher_labels is not None:      # test aid: pseudo-labels computed elsewhere (e.g. by an fp32 run, to compare numerics modes on equal labels)
                per_teacher.append(per_teacher_labels[len(per_teacher)])
            else:
                per_teacher.append(P.logits_to_ground_truth(logits_t, image_size, conf_threshold, nms_threshold,
                                                            inclusive=inclusive_nms))
        if kd_mode == "pairwise":
            kd.append(L.mta_loss(feats_s, feats_t, T, p))
        else:
            feats_all.append(feats_t)
    if kd_mode != "pairwise":
        kd.append(L.mta_loss(feats_s, feats_all, T, p))
    labels = P.merge_teacher_labels(per_teacher, B, 0.5, inclusive_nms, merge01=augment)
    reg, cls = L.focal_loss(logits_s[0], logits_s[1], logits_s[2], labels)
    return {"reg": reg, "cls": cls, "kd": kd, "labels": labels, "per_teacher": per_teacher,
            "logits_s": logits_s, "features_s": feats_s}


def total_loss(out, w_main: float = 1.0, w_kd: float = 0.005) -> torch.Tensor:
    """traditional.py:171-181: w_main*(mean(reg)+mean(cls)) + w_kd*sum(stack(kd))."""
    loss_main = torch.mean(torch.stack([out["reg"]])) + torch.mean(torch.stack([out["cls"]]))
    return w_main * loss_main + w_kd * torch.sum(torch.stack(out["kd"]))


def adam_step(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], state: Dict[str, Dict],
              lr: float = 1e-4, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam (no weight decay, no amsgrad), bias-corrected; parameters without a gradient
    are skipped exactly like the optimizer skips `p.grad is None`."""
    for k, pth in params.items():
        g = grads.get(k)
        if g is None:
            continue
        s = state.setdefault(k, {"step": 0, "m": torch.zeros_like(pth), "v": torch.zeros_like(pth)})
        s["step"] += 1
        s["m"].mul_(b1).add_(g, alpha=1 - b1)
        s["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** s["step"]
        bc2 = 1 - b2 ** s["step"]
        denom = (s["v"].sqrt() / (bc2 ** 0.5)).add_(eps)
        pth.addcdiv_(s["m"], denom, value=-(lr / bc1))
