"""CPU: drop-in surface plumbing — state-dict keys/shapes of the nn.Module facade, native-layout views, metrics."""
import json
import os

import numpy as np
import torch

from mm_distillnet_amd.arch import make_spec
from mm_distillnet_amd.metrics import ap_table, bbox_iou, compute_ap
from mm_distillnet_amd.model import YetAnotherEfficientDet, filter_state_dict
from mm_distillnet_amd.synth import synth_state


def test_facade_state_dict_matches_reference_keys(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "state_keys_d2_c8.json")))
    m = YetAnotherEfficientDet(compound_coef=2, in_channels=8, device="cpu")
    sd = m.state_dict()
    assert [[k, list(v.shape)] for k, v in sd.items()] == ref
    assert [k for k, _ in m.named_parameters()] == [k for k, s in ref if "running" not in k and "num_batches" not in k]
    st = synth_state(make_spec(2, 8), seed=5)
    m.load_state_dict(st)
    for k, v in m.state_dict().items():
        assert torch.equal(v.cpu().to(st[k].dtype), st[k]), k
    # parameters are views of the engine's flat buffer: an in-place optimizer-style update is seen by the engine
    p = dict(m.named_parameters())["backbone_net.model._blocks.3._depthwise_conv.conv.weight"]
    with torch.no_grad():
        p.add_(1.0)
    ex = m._net.ps.export_state()
    assert torch.allclose(ex["backbone_net.model._blocks.3._depthwise_conv.conv.weight"],
                          st["backbone_net.model._blocks.3._depthwise_conv.conv.weight"] + 1.0)


def test_filter_state_dict_remaps_module_and_generator_keys():
    st = synth_state(make_spec(2, 3), seed=1)
    keys = {k: v.shape for k, v in st.items()}
    weird = {}
    for k, v in st.items():
        if k.startswith("backbone_net"):
            weird["module." + k.replace("backbone_net", "model_backbones.rgb")] = v
        elif k.startswith("bifpn"):
            weird[k.replace("bifpn", "model_necks.rgb")] = v
        else:
            weird["module." + k] = v
    weird["regressor.header.pointwise_conv.conv.bias"] = torch.zeros(3)     # wrong shape: dropped
    out = filter_state_dict(keys, weird)
    assert set(out) == set(st)


def test_metrics_small_case():
    assert abs(bbox_iou(np.array([[0, 0, 9, 9]]), np.array([[0, 0, 9, 9]]))[0] - 1.0) < 1e-6
    assert abs(compute_ap(np.array([0.5, 1.0]), np.array([1.0, 0.5])) - 0.75) < 1e-9
    preds = [np.array([[0, 0, 10, 10, 0.9, 6], [50, 50, 60, 60, 0.8, 6]], np.float32), np.zeros((0, 6), np.float32)]
    labels = [np.array([[0, 0, 10, 10, 6]], np.float32), np.array([[5, 5, 9, 9, 6]], np.float32)]
    t = ap_table(preds, labels)
    # one TP then one FP, 2 ground-truth boxes -> recall 0.5 at precision 1 -> AP 0.5 at every IoU
    assert abs(t["AP@0.5"] - 50.0) < 1e-6 and abs(t["AP@Ave"] - 50.0) < 1e-6


def _ragged(rows, counts, cols):
    out, o = [], 0
    for c in counts:
        out.append(rows[o:o + c].reshape(-1, cols)); o += c
    return out


def test_metrics_reference_golden(golden_dir):
    """metrics.py against the reference's own get_batch_statistics / ap_per_class / get_batch_central_distances and the
    AP@Ave / AP@0.5 / AP@0.75 / CDx / CDy row its evaluate() wrote for the same synthetic detections
    (tests/golden/metrics_eval.npz, tools/oracle/make_golden.py metrics)."""
    import os
    from mm_distillnet_amd import metrics as M
    g = np.load(os.path.join(golden_dir, "metrics_eval.npz"))
    S = int(g["image_size"])
    preds = _ragged(g["pred_rows"], g["pred_counts"], 6)
    labs = _ragged(g["lab_rows"], g["lab_counts"], 5)
    bs = 4
    all_pred = [preds[i:i + bs] for i in range(0, len(preds), bs)]
    all_lab = [labs[i:i + bs] for i in range(0, len(labs), bs)]
    labels = [float(r[4]) for l in labs for r in l]
    for iou in (0.5, 0.75, 0.9):
        sm = []
        for bp, bl in zip(all_pred, all_lab):
            sm += M.get_batch_statistics(bp, bl, iou)
        tp, sc, lb = [np.concatenate(x, 0) for x in zip(*sm)]
        np.testing.assert_array_equal(tp, g[f"tp@{iou}"])
        np.testing.assert_array_equal(sc, g[f"score@{iou}"])
        np.testing.assert_array_equal(lb, g[f"label@{iou}"])
        p, r, ap, f1, cls, ratio = M.ap_per_class(tp, sc, lb, labels)
        np.testing.assert_array_equal(p, g[f"precision@{iou}"]); np.testing.assert_array_equal(r, g[f"recall@{iou}"])
        np.testing.assert_array_equal(ap, g[f"ap@{iou}"]); np.testing.assert_array_equal(f1, g[f"f1@{iou}"])
        np.testing.assert_array_equal(cls, g[f"ap_class@{iou}"])
        assert ratio == float(g[f"score_ratio@{iou}"])
    cdx, cdy = [], []
    for bp, bl in zip(all_pred, all_lab):
        x, y = M.get_batch_central_distances(bp, bl, S, S)
        cdx.extend(x); cdy.extend(y)
    np.testing.assert_array_equal(np.array(cdx, dtype=np.float64), g["cd_x"])
    np.testing.assert_array_equal(np.array(cdy, dtype=np.float64), g["cd_y"])
    table = M.evaluate_table(all_pred, all_lab, labels, S)
    for k in ("AP@Ave", "AP@0.5", "AP@0.75"):      # float64 through the csv's 15 printed digits
        assert abs(table[k] - float(g["table." + k])) <= 1e-12 * abs(table[k]), (k, table[k], float(g["table." + k]))
    for k in ("CDx", "CDy"):      # the csv round trip prints 7-8 significant digits of a float32 mean
        assert abs(table[k] - float(g["table." + k])) <= 1e-6 * abs(table[k]), (k, table[k], float(g["table." + k]))
    # the reference's sentinel row when nothing was predicted
    empty = M.evaluate_table([[np.zeros((0, 6), np.float32)]], [[labs[0] if len(labs[0]) else labs[2]]], [6.0], S)
    assert empty["AP@0.5"] == 0.0 and empty["CDx"] == 10000.0 and empty["CDy"] == 10000.0
