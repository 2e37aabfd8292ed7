"""CPU: drop-in surface plumbing — state-dict keys/shapes of the nn.Module facade, native-layout views, metrics."""
import json
import os

import numpy as np
import pytest
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


def _host_engine(optimizer="Adam", **kw):
    """DistillEngine-shaped host object (flat buffers on the CPU) for the checkpoint-format functions of trainer.py."""
    import types
    from mm_distillnet_amd.arch import make_spec
    from mm_distillnet_amd.store import ParamStore
    from mm_distillnet_amd.step import StepConfig, OPT_MODES
    spec = make_spec(2, 8)
    ps = ParamStore(spec, "cpu", with_grads=True)
    eng = types.SimpleNamespace()
    eng.cfg = StepConfig(optimizer=optimizer, **kw)
    eng.opt_mode = OPT_MODES[optimizer]
    eng.student = types.SimpleNamespace(ps=ps, spec=spec)
    eng.exp_avg, eng.exp_avg_sq = torch.zeros(ps.n_params), torch.zeros(ps.n_params)
    eng.adam_main, eng.adam_head = torch.zeros(4), torch.zeros(4)
    eng.head_active = torch.zeros(1, dtype=torch.int32)
    eng.hyper = torch.tensor([1e-4, 0.9, 0.999, 1e-8, 0.0, 0.0])
    eng.lr = 1e-4

    def set_lr(lr):
        eng.lr = float(lr); eng.hyper[0] = lr
    eng.set_lr = set_lr
    return eng, spec


def test_optimizer_state_is_torch_adam_state_dict(golden_dir):
    """Checkpoint `optimizer` payload = torch.optim.Adam.state_dict() over the reference's named_parameters() order
    (src/optimization/train_methods.py:1049-1064 saves optimizer.state_dict(); :1188-1236 loads it back).  A state_dict produced
    by a real torch Adam on reference-shaped parameters is imported, exported again and loaded into a fresh torch Adam: identical."""
    import json, os
    from mm_distillnet_amd import trainer as TR
    from mm_distillnet_amd.layout import param_rows
    eng, spec = _host_engine()
    rows = param_rows(spec)
    ref_keys = [k for k, shp in json.load(open(os.path.join(golden_dir, "state_keys_d2_c8.json")))
                if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
    assert [r[0] for r in rows] == ref_keys                 # == named_parameters() order of the reference model
    g = torch.Generator().manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(shape, generator=g) * 0.1) for _, shape, _ in rows]
    head = [k.startswith(("regressor", "classifier")) for k, _, _ in rows]
    opt = torch.optim.Adam(params, lr=3e-5, betas=(0.9, 0.999))
    for it in range(3):                                    # heads get gradients from the second step on (first batch had no labels)
        for p_, h in zip(params, head):
            p_.grad = None if (h and it == 0) else torch.randn(p_.shape, generator=g) * 0.01
        opt.step()
    sd = opt.state_dict()
    TR.load_optimizer_state_dict(eng, sd)
    assert eng.adam_main[0].item() == 3 and eng.adam_head[0].item() == 2 and eng.head_active.item() == 1 and eng.lr == 3e-5
    out = TR.optimizer_state_dict(eng)
    assert sorted(out["state"].keys()) == sorted(sd["state"].keys())
    for i, e in sd["state"].items():
        assert int(e["step"]) == out["state"][i]["step"]
        assert torch.equal(e["exp_avg"], out["state"][i]["exp_avg"]) and torch.equal(e["exp_avg_sq"], out["state"][i]["exp_avg_sq"])
    assert out["param_groups"][0]["params"] == sd["param_groups"][0]["params"] and out["param_groups"][0]["lr"] == 3e-5
    opt2 = torch.optim.Adam([torch.nn.Parameter(p_.detach().clone()) for p_ in params], lr=1.0)
    opt2.load_state_dict({k: v for k, v in out.items() if k != "mmd_steps"})        # torch accepts it as its own
    sd2 = opt2.state_dict()
    assert sd2["param_groups"][0]["lr"] == 3e-5
    for i, e in sd["state"].items():
        assert float(sd2["state"][i]["step"]) == float(e["step"]) and torch.equal(sd2["state"][i]["exp_avg"], e["exp_avg"])
    # before any labelled batch: head parameters have no state at all, like torch (grad is None -> skipped)
    eng2, _ = _host_engine()
    eng2.adam_main[0] = 1.0
    o2 = TR.optimizer_state_dict(eng2)
    assert len(o2["state"]) == sum(1 for h in head if not h)
    # SGD with momentum: momentum_buffer entries
    eng3, _ = _host_engine("SGD", momentum=0.9, weight_decay=1e-4)
    eng3.adam_main[0] = 2.0; eng3.exp_avg.fill_(0.5)
    o3 = TR.optimizer_state_dict(eng3)
    assert o3["param_groups"][0]["momentum"] == 0.9 and torch.all(o3["state"][0]["momentum_buffer"] == 0.5)
    eng4, _ = _host_engine("SGD", momentum=0.9)
    TR.load_optimizer_state_dict(eng4, o3)
    assert eng4.adam_main[0].item() == 2 and eng4.exp_avg[:100].eq(0.5).all()


def test_scheduler_surface_matches_torch():
    """cfg `scheduler` honoured or rejected like upstream (src/optimization/train_methods.py:860-878); state_dict is torch's."""
    import configparser
    from mm_distillnet_amd import trainer as TR
    eng, _ = _host_engine()
    c = configparser.ConfigParser()
    c["DEFAULT"] = {"lr": "1e-4", "scheduler": "ReduceLROnPlateau", "step_size": "2", "gamma": "0.5"}
    cfg = c["DEFAULT"]
    s = TR.LrSchedule(eng, cfg)
    ref_p = torch.nn.Parameter(torch.zeros(1))
    ref_o = torch.optim.Adam([ref_p], lr=1e-4)
    ref = torch.optim.lr_scheduler.ReduceLROnPlateau(ref_o, patience=3)
    for loss in [1.0, 0.9, 0.95, 0.96, 0.97, 0.98, 0.99, 1.0, 1.0, 1.0, 1.0]:
        s.step(loss); ref.step(loss)
        assert eng.lr == ref_o.param_groups[0]["lr"]
    assert abs(eng.lr - 1e-6) < 1e-18
    assert s.state_dict() == ref.state_dict()
    s2 = TR.LrSchedule(eng, cfg)
    s2.load_state_dict(ref.state_dict())
    assert s2.state_dict() == ref.state_dict()
    cfg["scheduler"] = "StepLR"
    eng.set_lr(1e-4)
    s3 = TR.LrSchedule(eng, cfg)
    for _ in range(4):
        s3.step(0.0)
    assert abs(eng.lr - 0.25e-4) < 1e-12
    cfg["scheduler"] = "CosineAnnealingWarmRestarts"
    eng.set_lr(1e-4)
    s4 = TR.LrSchedule(eng, cfg); s4.step(1.0)
    assert eng.lr == 1e-4                                   # constructed but never stepped upstream
    cfg["scheduler"] = "OneCycle"
    with pytest.raises(Exception, match="Unsupported scheduler OneCycle"):
        TR.LrSchedule(eng, cfg)
    cfg["optimizer"] = "RMSprop"
    with pytest.raises(Exception, match="Unsupported optimizer RMSprop"):
        TR.optimizer_settings(cfg)
