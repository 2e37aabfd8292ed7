"""ORACLE (test infrastructure only): CPU restatement of the two live losses of MM-DistillNet.

  mta_loss ........ src/loss/MTALoss.py:15-77  (pairwise mode and list-of-teachers mode)
  focal_loss ...... src/loss/YetAnotherFocalLoss.py:6-190 (calc_iou + focal + smooth-L1)

Written with differentiable torch-CPU ops so autograd yields the reference backward.  Quirks kept
on purpose: kl_div is fed probabilities (not log-probs) as its first argument; an
image without boxes in a batch that has boxes contributes an UN-normalised classification loss; a
batch without any box returns zeros.

Pinned by tests/golden/loss_*.npz (made from the reference's own classes by
tools/oracle/make_golden.py).
"""
from __future__ import annotations

from typing import List, Sequence, Union

import numpy as np
import torch
import torch.nn.functional as F


def attention_map(f: torch.Tensor, p: float) -> torch.Tensor:
    """at(f) = L2-normalise(mean_c f^p) over HW  (MTALoss.py:76-77). f: [B,C,H,W]."""
    return F.normalize(f.pow(p).mean(1).view(f.size(0), -1))


def mta_level(f_s: torch.Tensor, f_t: Union[torch.Tensor, Sequence[torch.Tensor]], T: float, p: float):
    a_s = attention_map(f_s, p)
    if torch.is_tensor(f_t):
        a_t = attention_map(f_t, p)
    elif len(f_t) == 1:
        a_t = attention_map(f_t[0], p)
    else:
        q = attention_map(f_t[0], p)
        for k in range(1, len(f_t)):
            q = q * attention_map(f_t[k], p)
        a_t = F.normalize(q, dim=1, p=1)
    u = F.softmax(a_s / T, dim=1)
    v = F.softmax(a_t / T, dim=1)
    # F.kl_div(input=u, target=v, 'batchmean') = sum(v*(log v - u))/B   (input is NOT log-prob here)
    return (torch.xlogy(v, v) - v * u).sum() / f_s.size(0)


def mta_loss(g_s, g_t, T: float = 9.0, p: float = 2.0) -> torch.Tensor:
    """g_s: 5 student maps. g_t: 5 teacher maps (pairwise) or list of per-teacher 5-lists (list mode)."""
    if torch.is_tensor(g_t[0]):
        return torch.stack([mta_level(s, t, T, p) for s, t in zip(g_s, g_t)])
    return torch.stack([mta_level(g_s[i], [ft[i] for ft in g_t], T, p) for i in range(len(g_s))])


def calc_iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a: anchors [A,4] (y1,x1,y2,x2); b: boxes [G,4] (x1,y1,x2,y2)  (YetAnotherFocalLoss.py:6-20)."""
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    iw = torch.min(a[:, 3:4], b[:, 2]) - torch.max(a[:, 1:2], b[:, 0])
    ih = torch.min(a[:, 2:3], b[:, 3]) - torch.max(a[:, 0:1], b[:, 1])
    iw = torch.clamp(iw, min=0)
    ih = torch.clamp(ih, min=0)
    ua = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])).unsqueeze(1) + area - iw * ih
    ua = torch.clamp(ua, min=1e-8)
    return iw * ih / ua


def focal_loss(classifications: torch.Tensor, regressions: torch.Tensor, anchors: torch.Tensor,
               annotations: List[np.ndarray]):
    """Returns (regression_loss[1], classification_loss[1]) like the reference (:27-190)."""
    alpha, gamma = 0.25, 2.0
    anchor = anchors[0]
    dtype = anchors.dtype
    max_n = max(int(np.shape(a)[0]) if np.ndim(a) > 0 else 0 for a in annotations)
    B = classifications.shape[0]
    aw = anchor[:, 3] - anchor[:, 1]
    ah = anchor[:, 2] - anchor[:, 0]
    acx = anchor[:, 1] + 0.5 * aw
    acy = anchor[:, 0] + 0.5 * ah
    cls_losses, reg_losses = [], []
    for j in range(B):
        if max_n == 0:
            continue   # padded annotation tensor has zero elements for every image
        ann = annotations[j]
        box = torch.from_numpy(np.asarray(ann, dtype=np.float32).reshape(-1, 5)) if np.size(ann) else torch.zeros(0, 5)
        c = torch.clamp(classifications[j], 1e-4, 1.0 - 1e-4)
        r = regressions[j]
        if box.shape[0] == 0:
            cls_losses.append(((1.0 - alpha) * c.pow(gamma) * (-torch.log(1.0 - c))).sum())
            reg_losses.append(torch.tensor(0, dtype=dtype))
            continue
        iou = calc_iou(anchor, box[:, :4])
        iou_max, iou_arg = torch.max(iou, dim=1)
        targets = torch.ones_like(c) * -1
        targets[iou_max < 0.4, :] = 0
        pos = iou_max >= 0.5
        npos = pos.sum()
        assigned = box[iou_arg, :]
        targets[pos, :] = 0
        targets[pos, assigned[pos, 4].long()] = 1
        af = torch.where(targets == 1., torch.full_like(c, alpha), torch.full_like(c, 1 - alpha))
        fw = torch.where(targets == 1., 1. - c, c)
        fw = af * fw.pow(gamma)
        bce = -(targets * torch.log(c) + (1.0 - targets) * torch.log(1.0 - c))
        cl = torch.where(targets != -1.0, fw * bce, torch.zeros_like(c))
        cls_losses.append(cl.sum() / torch.clamp(npos.to(dtype), min=1.0))
        if npos > 0:
            asg = assigned[pos]
            gw = asg[:, 2] - asg[:, 0]
            gh = asg[:, 3] - asg[:, 1]
            gcx = asg[:, 0] + 0.5 * gw
            gcy = asg[:, 1] + 0.5 * gh
            gw = torch.clamp(gw, min=1)
            gh = torch.clamp(gh, min=1)
            t = torch.stack(((gcy - acy[pos]) / ah[pos], (gcx - acx[pos]) / aw[pos],
                             torch.log(gh / ah[pos]), torch.log(gw / aw[pos]))).t()
            d = torch.abs(t - r[pos])
            reg_losses.append(torch.where(d <= 1.0 / 9.0, 0.5 * 9.0 * d.pow(2), d - 0.5 / 9.0).mean())
        else:
            reg_losses.append(torch.tensor(0, dtype=dtype))
    cls = torch.stack(cls_losses).mean(dim=0, keepdim=True) if cls_losses else torch.zeros(1)
    reg = torch.stack(reg_losses).mean(dim=0, keepdim=True) if reg_losses else torch.zeros(1)
    return reg, cls
