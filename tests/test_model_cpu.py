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
