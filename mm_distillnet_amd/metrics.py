"""Detection metrics of the reference's evaluate(), numpy restatement on the host:
  bbox_iou (+1 pixel convention) ....... src/utils/utils.py:1139-1185
  get_batch_statistics ................. src/utils/utils.py:1058-1136
  ap_per_class / compute_ap ............ src/utils/utils.py:1188-1280
  AP@0.5 / AP@0.75 / AP@Ave table ...... src/utils/utils.py:2096-2181 (IoU in np.arange(0.5, 0.95, 0.05))
These run once per evaluation, not per step; they consume the device-side predictions / pseudo-labels."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


def bbox_iou(box1: np.ndarray, box2: np.ndarray) -> np.ndarray:
    b1, b2 = np.asarray(box1, np.float32).reshape(-1, 4), np.asarray(box2, np.float32).reshape(-1, 4)
    ix1 = np.maximum(b1[:, 0], b2[:, 0]); iy1 = np.maximum(b1[:, 1], b2[:, 1])
    ix2 = np.minimum(b1[:, 2], b2[:, 2]); iy2 = np.minimum(b1[:, 3], b2[:, 3])
    inter = np.clip(ix2 - ix1 + 1, 0, None) * np.clip(iy2 - iy1 + 1, 0, None)
    a1 = (b1[:, 2] - b1[:, 0] + 1) * (b1[:, 3] - b1[:, 1] + 1)
    a2 = (b2[:, 2] - b2[:, 0] + 1) * (b2[:, 3] - b2[:, 1] + 1)
    return inter / (a1 + a2 - inter + 1e-16)


def get_batch_statistics(outputs: Sequence, targets: Sequence, iou_threshold: float):
    """outputs[i]: [n,6] (x1,y1,x2,y2,score,label); targets[i]: [m,5] (x1,y1,x2,y2,label)."""
    metrics = []
    for out, tgt in zip(outputs, targets):
        out = np.asarray(out, np.float32).reshape(-1, 6) if np.size(out) else np.zeros((0, 6), np.float32)
        tgt = np.asarray(tgt, np.float32).reshape(-1, 5) if np.size(tgt) else np.zeros((0, 5), np.float32)
        if len(out) < 1 or len(tgt) < 1:
            continue
        tp = np.zeros(out.shape[0])
        detected = []
        for pi, row in enumerate(out):
            if len(detected) == len(tgt):
                break
            if row[5] not in tgt[:, 4]:
                continue
            ious = bbox_iou(row[None, :4], tgt[:, :4])
            bi = int(np.argmax(ious))
            if ious[bi] >= iou_threshold and bi not in detected:
                tp[pi] = 1
                detected.append(bi)
        metrics.append([tp, out[:, 4], out[:, 5]])
    return metrics


def compute_ap(recall, precision):
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([0.0], precision, [0.0]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def ap_per_class(tp, conf, pred_cls, target_cls):
    i = np.argsort(-conf)
    tp, conf, pred_cls = tp[i], conf[i], pred_cls[i]
    ap, p, r = [], [], []
    for c in np.unique(target_cls):
        i = pred_cls == c
        n_gt, n_p = (target_cls == c).sum(), i.sum()
        if n_p == 0 and n_gt == 0:
            continue
        if n_p == 0 or n_gt == 0:
            ap.append(0); r.append(0); p.append(0)
            continue
        fpc, tpc = (1 - tp[i]).cumsum(), tp[i].cumsum()
        rc = tpc / (n_gt + 1e-16)
        pc = tpc / (tpc + fpc)
        r.append(rc[-1]); p.append(pc[-1]); ap.append(compute_ap(rc, pc))
    return np.array(p), np.array(r), np.array(ap)


def ap_table(all_predictions: List, all_labels: List) -> dict:
    """all_predictions / all_labels: lists over images.  Returns AP@0.5, AP@0.75, AP@Ave in percent."""
    labels = np.concatenate([np.asarray(t, np.float32).reshape(-1, 5)[:, 4] for t in all_labels if np.size(t)]) \
        if any(np.size(t) for t in all_labels) else np.zeros((0,))
    rec, out = [], {"AP@0.5": 0.0, "AP@0.75": 0.0}
    for iou in np.arange(0.5, 0.95, 0.05):
        iou = float(np.around(iou, decimals=2))
        sm = get_batch_statistics(all_predictions, all_labels, iou)
        mean = 0.0
        if sm:
            tp, sc, lb = [np.concatenate(x, 0) for x in zip(*sm)]
            _, _, ap = ap_per_class(tp, sc, lb, labels)
            mean = float(ap.mean()) if ap.size else 0.0
        if iou == 0.5:
            out["AP@0.5"] = mean * 100
        if iou == 0.75:
            out["AP@0.75"] = mean * 100
        rec.append(mean)
    out["AP@Ave"] = float(np.mean(rec)) * 100
    return out
