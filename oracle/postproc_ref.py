"""ORACLE (test infrastructure only): pseudo-label generation on the CPU, numpy/torch restatement.

  decode_boxes ............... src/YetAnotherEfficientDet.py:574-602 (YetAnotherEfficientDetBBoxTransform)
  clip ....................... src/utils/utils.py:123-141 (ClipBoxes, clamps to image_size, not size-1)
  post_process ............... src/utils/utils.py:144-231 (EfficientDet_post_processing)
  logits_to_ground_truth ..... src/utils/utils.py:234-324 (int() truncation, label remap)
  merge_teacher_labels ....... src/optimization/train_methods.py:361-411 (concat per image + nms 0.5)
  nms / batched_nms .......... torchvision==0.4.2 (requirements.txt:322; NOT vendored in the reference):
        greedy NMS over boxes sorted by descending score, IoU = inter/(areaA+areaB-inter) with no +1,
        suppress when IoU > thr, kept indices returned in score order; batched_nms adds
        idx*(max_coordinate+1) to every box first.  `inclusive=True` switches to IoU >= thr, which is
        what torchvision's CPU kernel did before its CPU/CUDA consistency fix; the CUDA kernel has
        always used '>'.  Parity for this third-party piece is pinned only through the reference's
        call sites driven with the same restated NMS (tools/oracle/refshim.py) — "parity unpinned" at
        the torchvision boundary itself.
All arithmetic in float32 to mirror the tensors the reference feeds to torchvision.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

CAR_VALID = {"labels_txt2i": {"car": 6}, "labels_i2txt": {6: "car"},
             "predictions_txt2i": {"car": 6}, "predictions_i2txt": {6: "car"}}


def nms(boxes: np.ndarray, scores: np.ndarray, thr: float, inclusive: bool = False) -> np.ndarray:
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=np.float32).reshape(-1)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = ((x2 - x1) * (y2 - y1)).astype(np.float32)
    order = np.argsort(-scores, kind="stable")
    sup = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(thr)
    for _i in range(n):
        i = order[_i]
        if sup[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        if rest.size == 0:
            break
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), (xx2 - xx1).astype(np.float32))
        h = np.maximum(np.float32(0), (yy2 - yy1).astype(np.float32))
        inter = (w * h).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = (inter / ((areas[i] + areas[rest]).astype(np.float32) - inter)).astype(np.float32)
        hit = (ovr >= thr) if inclusive else (ovr > thr)
        sup[rest[hit]] = True
    return np.asarray(keep, dtype=np.int64)


def batched_nms(boxes, scores, idxs, thr, inclusive=False):
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    if boxes.shape[0] == 0:
        return np.zeros((0,), dtype=np.int64)
    maxc = boxes.max()
    off = (np.asarray(idxs).astype(np.float32) * (maxc + np.float32(1))).astype(np.float32)
    return nms((boxes + off[:, None]).astype(np.float32), scores, thr, inclusive)


def decode_boxes(anchors: torch.Tensor, regression: torch.Tensor) -> torch.Tensor:
    yca = (anchors[..., 0] + anchors[..., 2]) / 2
    xca = (anchors[..., 1] + anchors[..., 3]) / 2
    ha = anchors[..., 2] - anchors[..., 0]
    wa = anchors[..., 3] - anchors[..., 1]
    w = regression[..., 3].exp() * wa
    h = regression[..., 2].exp() * ha
    yc = regression[..., 0] * ha + yca
    xc = regression[..., 1] * wa + xca
    return torch.stack([xc - w / 2., yc - h / 2., xc + w / 2., yc + h / 2.], dim=2)


def post_process(classification, regression, anchors, image_size: int, conf_threshold: float,
                 nms_threshold: float, valid_prediction_ids: Sequence[int], inclusive=False) -> List[np.ndarray]:
    """-> per image float32 [n,6] rows (x1,y1,x2,y2,score,class_id), in NMS keep order."""
    boxes = decode_boxes(anchors[[0]], regression).clone()
    boxes[:, :, 0] = torch.clamp(boxes[:, :, 0], min=0)
    boxes[:, :, 1] = torch.clamp(boxes[:, :, 1], min=0)
    boxes[:, :, 2] = torch.clamp(boxes[:, :, 2], max=image_size)
    boxes[:, :, 3] = torch.clamp(boxes[:, :, 3], max=image_size)
    scores, classes = torch.max(classification, dim=2)
    out = []
    valid = torch.tensor(list(valid_prediction_ids), dtype=classes.dtype)
    for i in range(classification.shape[0]):
        m = scores[i] > conf_threshold
        if m.sum() == 0:
            out.append(np.zeros((0, 6), dtype=np.float32))
            continue
        b, s, c = boxes[i, m], scores[i, m], classes[i, m]
        s_unfiltered = s
        vm = (c[:, None] == valid[None, :]).any(-1)
        b, s, c = b[vm], s[vm], c[vm]
        keep = batched_nms(b.numpy(), s.numpy(), c.numpy(), nms_threshold, inclusive)
        if keep.shape[0] == 0:
            out.append(np.zeros((0, 6), dtype=np.float32))
            continue
        # REFERENCE QUIRK (src/utils/utils.py:193-213): `scores_` is never class-filtered, yet it is
        # indexed with the NMS indices of the class-filtered list, so the emitted score column is the
        # score of the idx-th OVER-THRESHOLD candidate (anchor order), not of the kept box.  That
        # column then orders the cross-teacher NMS, so it is preserved bit for bit.
        out.append(np.hstack((b.numpy()[keep], s_unfiltered.numpy()[keep].reshape(-1, 1),
                              c.numpy()[keep].reshape(-1, 1).astype(np.float32))).astype(np.float32))
    return out


def logits_to_ground_truth(logits, image_size: int, conf_threshold: float, nms_threshold: float,
                           valid: Dict = CAR_VALID, include_scores: bool = True, inclusive=False) -> List[np.ndarray]:
    """-> per image float32 [n,6] (x1,y1,x2,y2,score,label) with int()-truncated coords; empty -> shape (0,)."""
    cls, reg, anc = logits
    preds = post_process(cls, reg, anc, image_size, conf_threshold, nms_threshold,
                         list(valid["predictions_txt2i"].values()), inclusive)
    gts = []
    for p in preds:
        rows = []
        for r in p.tolist():
            x1 = int(max(r[0], 0)); y1 = int(max(r[1], 0))
            x2 = int(min(r[2], image_size)); y2 = int(min(r[3], image_size))
            label = valid["labels_txt2i"][valid["predictions_i2txt"][int(r[5])]]
            rows.append([x1, y1, x2, y2, r[4], label] if include_scores else [x1, y1, x2, y2, label])
        gts.append(np.array(rows, dtype=np.float32))
    return gts


def merge_teacher_labels(per_teacher: List[List[np.ndarray]], batch: int, iou: float = 0.5,
                         inclusive=False, merge01=False) -> List[np.ndarray]:
    """Concat each image's [n,6] rows over teachers (teacher order preserved), class-agnostic NMS at 0.5 on
    the int-truncated boxes, drop the score column, reorder by NMS keep order -> [m,5]; empty -> []."""
    merged: List = [[] for _ in range(batch)]
    for labels in per_teacher:
        for i in range(batch):
            a = labels[i]
            if np.size(a) == 0:
                continue
            a = a.reshape(-1, 6)
            merged[i] = a if np.size(merged[i]) == 0 else np.concatenate((merged[i], a), axis=0)
    # augment=True (src/optimization/train_methods.py:379-387): image 1 also gets image 0's rows, in front, if both have any
    if merge01 and batch >= 2 and np.size(merged[0]) and np.size(merged[1]):
        merged[1] = np.concatenate((merged[0], merged[1]), axis=0)
    out = []
    for i in range(batch):
        if np.size(merged[i]) == 0:
            out.append([])
            continue
        keep = nms(merged[i][:, 0:4], merged[i][:, 4], iou, inclusive)
        out.append(np.delete(merged[i], 4, 1)[keep])
    return out
