"""Detection metrics of the reference's evaluate(), numpy restatement on the host:
  bbox_iou (+1 pixel convention) ....... src/utils/utils.py:1139-1185
  get_batch_statistics ................. src/utils/utils.py:1058-1136
  get_batch_central_distances .......... src/utils/utils.py:979-1055 (CDx / CDy of arXiv 1910.11760 as the reference computes them)
  ap_per_class / compute_ap ............ src/utils/utils.py:1188-1280
  AP@0.5 / AP@0.75 / AP@Ave / CD table . src/utils/utils.py:2096-2181 (IoU in np.arange(0.5, 0.95, 0.05))
Pinned by tests/golden/metrics_eval.npz, produced by the reference's own functions and its evaluate() on a synthetic detection set
(tools/oracle/make_golden.py metrics).  These run once per evaluation, not per step; they consume the device-side predictions /
pseudo-labels."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


def bbox_iou(box1: np.ndarray, box2: np.ndarray) -> np.ndarray:
    """fp32 like the reference's torch tensors; the 1e-16 in the denominator is absorbed by fp32 rounding there too."""
    b1, b2 = np.asarray(box1, np.float32).reshape(-1, 4), np.asarray(box2, np.float32).reshape(-1, 4)
    ix1 = np.maximum(b1[:, 0], b2[:, 0]); iy1 = np.maximum(b1[:, 1], b2[:, 1])
    ix2 = np.minimum(b1[:, 2], b2[:, 2]); iy2 = np.minimum(b1[:, 3], b2[:, 3])
    one = np.float32(1)
    inter = np.clip(ix2 - ix1 + one, 0, None) * np.clip(iy2 - iy1 + one, 0, None)
    a1 = (b1[:, 2] - b1[:, 0] + one) * (b1[:, 3] - b1[:, 1] + one)
    a2 = (b2[:, 2] - b2[:, 0] + one) * (b2[:, 3] - b2[:, 1] + one)
    return (inter / (a1 + a2 - inter + np.float32(1e-16))).astype(np.float32)


def _rows(a, cols):
    return np.asarray(a, np.float32).reshape(-1, cols) if np.size(a) else np.zeros((0, cols), np.float32)


def get_batch_statistics(outputs: Sequence, targets: Sequence, iou_threshold: float):
    """outputs[i]: [n,6] (x1,y1,x2,y2,score,label); targets[i]: [m,5] (x1,y1,x2,y2,label) -> per image with both non-empty:
    [true_positives, scores, labels].  Predictions are visited in the given order (NMS keep order = descending score)."""
    metrics = []
    thr = np.float32(iou_threshold)           # torch compares the fp32 IoU tensor with the scalar in fp32
    for out, tgt in zip(outputs, targets):
        out, tgt = _rows(out, 6), _rows(tgt, 5)
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
            if ious[bi] >= thr and bi not in detected:
                tp[pi] = 1
                detected.append(bi)
        metrics.append([tp, out[:, 4], out[:, 5]])
    return metrics


def get_batch_central_distances(outputs: Sequence, targets: Sequence, width: float, height: float):
    """Per image with ground truth: mean |extent difference| between every ground-truth box and the closest not-yet-used
    prediction of its class, divided by the image size.  As in the reference the compared "points" are the boxes' EXTENTS
    (x2 - x1, y2 - y1), a ground truth without a same-class prediction counts with its full extent, and an image without
    predictions is compared against zeros."""
    cd_x, cd_y = [], []
    for out, tgt in zip(outputs, targets):
        tgt = _rows(tgt, 5)
        if len(tgt) < 1:
            continue
        tpt = tgt[:, 2:4] - tgt[:, 0:2]
        tlab = tgt[:, -1]
        out = _rows(out, 6)
        if len(out) < 1:
            plab = np.zeros_like(tlab)
            opt = np.zeros_like(tpt)
        else:
            plab = out[:, -1].copy()
            opt = out[:, 2:4] - out[:, 0:2]
        dx, dy = [], []
        for i in range(len(tpt)):
            sel = plab == tlab[i]
            valid = opt[sel]
            orig = np.arange(len(plab))[sel]
            if len(valid) < 1:
                dx.append(tpt[i, 0]); dy.append(tpt[i, 1])
            else:
                j = int(np.argmin(np.sum((valid - tpt[i]) ** 2, axis=1)))
                plab[orig[j]] = -1
                dx.append(np.abs(tpt[i, 0] - valid[j, 0])); dy.append(np.abs(tpt[i, 1] - valid[j, 1]))
        cd_x.append(np.mean(dx) / width)
        cd_y.append(np.mean(dy) / height)
    return cd_x, cd_y


def compute_ap(recall, precision):
    """Area under the precision envelope over the recall steps (the py-faster-rcnn rule the reference uses,
    src/utils/utils.py:1255-1280): precision is replaced by its running maximum from the right, and every change of recall
    contributes (delta recall) x (envelope right of the step)."""
    rec = np.r_[0.0, recall, 1.0]
    env = np.maximum.accumulate(np.r_[0.0, precision, 0.0][::-1])[::-1]
    width = np.diff(rec)
    steps = np.flatnonzero(rec[1:] != rec[:-1])
    return np.sum(width[steps] * env[steps + 1])


def ap_per_class(tp, conf, pred_cls, target_cls):
    """-> (precision, recall, AP, F1, classes, total_predictions / total_ground_truth), one entry per ground-truth class
    (src/utils/utils.py:1188-1252).  Classes come from the ground truth, so each has at least one object; a class nobody
    predicted scores 0 on all three."""
    target_cls = np.asarray(target_cls)
    ranked = np.argsort(-conf)
    hit, cls_ranked = tp[ranked], pred_cls[ranked]
    classes, n_objects = np.unique(target_cls, return_counts=True)
    prec, rec, ap = [], [], []
    n_pred = 0.0
    for c, n_gt in zip(classes, n_objects):
        mine = hit[cls_ranked == c]
        n_pred += mine.size
        if mine.size == 0:
            prec.append(0); rec.append(0); ap.append(0)
            continue
        hits, misses = np.cumsum(mine), np.cumsum(1 - mine)
        r_curve = hits / (n_gt + 1e-16)
        p_curve = hits / (hits + misses)
        prec.append(p_curve[-1]); rec.append(r_curve[-1]); ap.append(compute_ap(r_curve, p_curve))
    prec, rec, ap = np.array(prec), np.array(rec), np.array(ap)
    f1 = 2 * prec * rec / (prec + rec + 1e-16)
    return prec, rec, ap, f1, classes.astype("int32"), n_pred / float(n_objects.sum())


def evaluate_table(all_predictions: List[List], all_labels: List[List], labels: Sequence, image_size: int) -> dict:
    """The AP / CD row evaluate() writes (src/utils/utils.py:2096-2181).  all_predictions / all_labels: lists over batches of
    lists over images; labels: the flat list of ground-truth class ids.  Values in percent.  No prediction matched any image
    with ground truth at IoU 0.5 -> CDx = CDy = 100 * 100 (the reference's sentinel)."""
    out = {"AP@Ave": 0.0, "AP@0.5": 0.0, "AP@0.75": 0.0, "CDx": 0.0, "CDy": 0.0}
    rec = []
    for iou in np.arange(0.5, 0.95, 0.05):
        iou = np.around(iou, decimals=2)
        sm, cd_x, cd_y = [], [], []
        for bp, bl in zip(all_predictions, all_labels):
            sm += get_batch_statistics(bp, bl, iou)
            cx, cy = get_batch_central_distances(bp, bl, image_size, image_size)
            cd_x.extend(cx); cd_y.extend(cy)
        if not any(len(m[0]) for m in sm):      # `not any(sample_metrics)` upstream
            ap, cd_x, cd_y = [0.0], [100.0], [100.0]
            mean = 0.0
        else:
            tp, sc, lb = [np.concatenate(x, 0) for x in zip(*sm)]
            _, _, ap, _, _, _ = ap_per_class(tp, sc, lb, labels)
            mean = float(ap.mean()) if ap.size else float("nan")
        if iou == 0.5:
            out["AP@0.5"] = mean * 100
            out["CDx"] = float(np.mean(cd_x)) * 100
            out["CDy"] = float(np.mean(cd_y)) * 100
        if iou == 0.75:
            out["AP@0.75"] = mean * 100
        rec.append(mean)
    out["AP@Ave"] = float(np.mean(rec)) * 100
    return out


def ap_table(all_predictions: List, all_labels: List, image_size: int = 512) -> dict:
    """Flat per-image lists -> evaluate_table (one batch)."""
    labels = [float(r[4]) for t in all_labels for r in _rows(t, 5)]
    return evaluate_table([all_predictions], [all_labels], labels, image_size)
