"""CPU: the oracle/ restatement against the golden vectors captured from the reference."""
import json
import os

import numpy as np
import pytest
import torch

from mm_distillnet_amd.arch import make_spec
from mm_distillnet_amd.layout import state_layout
from mm_distillnet_amd.synth import synth_inputs
from oracle import effdet_ref as O
from oracle import losses_ref as L
from oracle import postproc_ref as P
from oracle import step_ref as ST
from helpers import make_state, check_summary, grad_state


@pytest.mark.parametrize("coef", [2, 4])
def test_state_layout_matches_reference_keys(golden_dir, coef):
    ref = json.load(open(os.path.join(golden_dir, f"state_keys_d{coef}_c8.json")))
    mine = [[k, list(s)] for k, s, _ in state_layout(make_spec(coef, 8))]
    assert mine == ref


@pytest.mark.parametrize("mod,cin,seed", [("rgb", 3, 11), ("thermal", 1, 12), ("audio", 8, 13)])
def test_net_eval(golden_dir, mod, cin, seed):
    g = np.load(os.path.join(golden_dir, f"net_d2_eval_{mod}.npz"))
    spec, st = make_state(2, cin, seed, mod)
    x = synth_inputs(2, 128, seed=24)[mod]
    with torch.no_grad():
        (c, r, a), f = O.forward(st, x, 2, False)
    check_summary(g, "cls", c); check_summary(g, "reg", r); check_summary(g, "anchors", a, 1e-6, 1e-7)
    for i, u in enumerate(f):
        check_summary(g, f"feat{i}", u)


def test_anchors_512(golden_dir):
    g = np.load(os.path.join(golden_dir, "anchors_d2_512.npz"))
    a = O.anchors_for(512, 2)
    assert a.shape == (1, 49104, 4)
    check_summary(g, "anchors512", a, 1e-7, 1e-7)
    np.testing.assert_array_equal(a[0, ::997].numpy(), g["anchors512.sample"])


def test_net_train_fwd_bwd(golden_dir):
    g = np.load(os.path.join(golden_dir, "net_d2_train_audio.npz"))
    spec, st = make_state(2, 8, 13, "audio")
    st = grad_state(st)
    x = synth_inputs(2, 128, seed=25)["audio"]
    masks = {int(b): torch.from_numpy(m) for b, m in zip(g["drop_blocks"], g["drop_masks"])}
    (c, r, a), f = O.forward(st, x, 2, True, masks)
    loss = c.sum() * 0.01 + (r ** 2).mean() + sum((u ** 2).mean() for u in f)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    check_summary(g, "cls", c); check_summary(g, "reg", r)
    for k in g.files:
        if k.startswith("grad.") and k.endswith(".head"):
            name = k[5:-5]
            check_summary(g, "grad." + name, st[name].grad, 2e-3, 1e-4)
        if k.startswith("stat.") and k.endswith(".head"):
            name = k[5:-5]
            check_summary(g, "stat." + name, st[name], 1e-5, 1e-6)
    assert int(st["backbone_net.model._bn0.num_batches_tracked"]) == int(g["nbt"])


def _focal_inputs(seed, B, A):
    gen = torch.Generator().manual_seed(seed)
    cls = torch.sigmoid(torch.randn(B, A, 20, generator=gen) * 2 - 2)
    cls[0, :50] = 0.0
    cls[1, :50] = 1.0
    reg = torch.randn(B, A, 4, generator=gen) * 0.3
    return cls, reg


@pytest.mark.parametrize("name", ["mixed", "all_empty", "ignore_band"])
def test_focal(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"loss_focal_{name}.npz"))
    S = int(g["image_size"])
    anchors = O.anchors_for(S, 2)
    ann = [g[f"ann{i}"] for i in range(3)]
    cls, reg = _focal_inputs(int(g["seed"]), 3, anchors.shape[1])
    check_summary(g, "cls_in", cls, 1e-6, 1e-7)
    cls.requires_grad_(True); reg.requires_grad_(True)
    rl, cl = L.focal_loss(cls, reg, anchors, ann)
    np.testing.assert_allclose(rl.detach().numpy(), g["reg_loss"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(cl.detach().numpy(), g["cls_loss"], rtol=1e-5, atol=1e-7)
    if "dcls.sample" in g.files:
        (rl.sum() + cl.sum()).backward()
        np.testing.assert_allclose(cls.grad.reshape(-1)[::211].numpy(), g["dcls.sample"], rtol=1e-4, atol=1e-8)
        np.testing.assert_allclose(reg.grad.reshape(-1)[::53].numpy(), g["dreg.sample"], rtol=1e-4, atol=1e-8)
        check_summary(g, "dcls", cls.grad); check_summary(g, "dreg", reg.grad)


@pytest.mark.parametrize("name", ["stock", "peaky"])
def test_mta(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"loss_mta_{name}.npz"))
    T = float(g["T"])
    fs = [torch.from_numpy(g[f"fs{i}"]).requires_grad_(True) for i in range(5)]
    fts = [[torch.from_numpy(g[f"ft{k}_{i}"]) for i in range(5)] for k in range(3)]
    pair = L.mta_loss(fs, fts[0], T, 2.0)
    np.testing.assert_allclose(pair.detach().numpy(), g["pair"], rtol=1e-5, atol=1e-6)
    pair.sum().backward()
    for i, f in enumerate(fs):
        np.testing.assert_allclose(f.grad.numpy(), g[f"pair_dfs{i}"], rtol=1e-3, atol=1e-9)
        f.grad = None
    lst = L.mta_loss(fs, fts, T, 2.0)
    np.testing.assert_allclose(lst.detach().numpy(), g["list"], rtol=1e-5, atol=1e-6)
    lst.sum().backward()
    for i, f in enumerate(fs):
        np.testing.assert_allclose(f.grad.numpy(), g[f"list_dfs{i}"], rtol=1e-3, atol=1e-9)


def _postproc_inputs(seed, B, A):
    gen = torch.Generator().manual_seed(seed)
    cls = torch.sigmoid(torch.randn(B, A, 20, generator=gen) * 1.5 - 3.0)
    cls[:, :, 6] = torch.sigmoid(torch.randn(B, A, generator=gen) * 2.0 - 1.5)
    cls[2] = cls[2] * 0.2
    reg = torch.randn(B, A, 4, generator=gen) * 0.4
    return cls, reg


def test_postproc(golden_dir):
    g = np.load(os.path.join(golden_dir, "postproc_d2_128.npz"))
    S = int(g["image_size"])
    anchors = O.anchors_for(S, 2)
    cls, reg = _postproc_inputs(int(g["seed"]), 3, anchors.shape[1])
    check_summary(g, "cls_in", cls, 1e-6, 1e-7)
    gts = P.logits_to_ground_truth([cls, reg, anchors], S, 0.3, 0.5)
    for i, gt in enumerate(gts):
        np.testing.assert_array_equal(gt.reshape(-1, 6), g[f"gt{i}"])


@pytest.mark.parametrize("variant", ["pairwise", "list", "augmented", "rgb1", "listaug"])
def test_step(golden_dir, variant):
    g = np.load(os.path.join(golden_dir, f"step_d2_256_{variant}.npz"))
    S, B = 256, 2
    bias = {"rgb": -2.0, "depth": -3.2, "thermal": -2.0}
    mods = {"rgb": (3, 21), "depth": (3, 22), "thermal": (1, 23)}
    if variant == "rgb1":      # BASELINE configs[1]: student + the RGB teacher only
        mods = {"rgb": mods["rgb"]}
    teachers = {k: make_state(2, cin, seed, k, cls_bias=bias[k])[1] for k, (cin, seed) in mods.items()}
    _, st = make_state(2, 8, 24, "audio")
    st = grad_state(st)
    batch = synth_inputs(B, S, seed=31)
    masks = {int(b): torch.from_numpy(m) for b, m in zip(g["drop_blocks"], g["drop_masks"])}
    # "augmented" = ModelWithNMSLossAugmented.forward(..., augment=True): pairwise KD + audio merge / feature averaging / label merge
    # "listaug" = ModelWithNMSKDListLossAugmented.forward(label=<RGB frames of other recordings>, augment=True)
    aug_rgb = synth_inputs(B, S, seed=57)["rgb"] if variant == "listaug" else None
    out = ST.distill_forward(st, teachers, batch, S, 2, masks, kd_mode="list" if variant in ("list", "listaug") else "pairwise",
                             augment=variant == "augmented", aug_rgb=aug_rgb)
    for ti in range(len(mods) + (1 if variant == "listaug" else 0)):
        for i in range(B):
            np.testing.assert_array_equal(out["per_teacher"][ti][i].reshape(-1, 6), g[f"teacher{ti}_img{i}"])
    np.testing.assert_allclose(out["reg"].detach().numpy(), g["reg"], rtol=1e-4)
    np.testing.assert_allclose(out["cls"].detach().numpy(), g["cls"], rtol=1e-4)
    np.testing.assert_allclose(torch.stack(out["kd"]).detach().numpy(), g["kd"], rtol=1e-5)
    loss = ST.total_loss(out)
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    loss.backward()
    params = {k: v for k, v in st.items() if v.requires_grad}
    grads = {k: v.grad for k, v in params.items() if v.grad is not None}
    for k in g.files:
        if k.startswith("grad.") and k.endswith(".head"):
            name = k[5:-5]
            check_summary(g, "grad." + name, grads[name], 2e-3, 1e-4)
    with torch.no_grad():
        ST.adam_step(params, grads, {})
    for k in g.files:
        if k.startswith("adam.") and k.endswith(".head"):
            name = k[5:-5]
            check_summary(g, "adam." + name, params[name], 1e-5, 1e-6)


VAL_MODS = {"rgb": (3, 21), "depth": (3, 22), "thermal": (1, 23)}
VAL_BIAS = {"rgb": -2.0, "depth": -3.2, "thermal": -2.0}


def val_states():
    """tools/oracle/make_golden.py val_states(), repeated with the shared recipe."""
    tstates = {k: make_state(2, cin, seed, k, cls_bias=VAL_BIAS[k])[1] for k, (cin, seed) in VAL_MODS.items()}
    spec, st_s = make_state(2, 8, 24, "audio", cls_bias=-2.0)
    return tstates, spec, st_s


def test_validate_and_predictions_golden(golden_dir):
    """The oracle's validation path (eval-mode student, no augmentation, no backward) against the reference's own validate() and
    get_predictions_multiteacher run on a 4-sample synthetic set (tests/golden/validate_d2_256.npz): per-batch loss sums, the val_loss
    / Test scalars assembled the way validate() does, student detections and merged multi-teacher labels row for row."""
    g = np.load(os.path.join(golden_dir, "validate_d2_256.npz"))
    S, N, B = int(g["image_size"]), int(g["n"]), int(g["batch"])
    tstates, spec, st_s = val_states()
    data = synth_inputs(N, S, seed=61)
    tot = reg_t = cls_t = kd_t = 0.0
    for b in range(N // B):
        batch = {k: v[b * B:(b + 1) * B] for k, v in data.items()}
        with torch.no_grad():
            out = ST.distill_forward(st_s, tstates, batch, S, 2, None, training=False)
        reg, cls, kd = out["reg"].item(), out["cls"].item(), torch.stack(out["kd"]).numpy()
        np.testing.assert_allclose(reg, g["batch_reg"][b], rtol=1e-5)
        np.testing.assert_allclose(cls, g["batch_cls"][b], rtol=1e-5)
        np.testing.assert_allclose(kd, g["batch_kd"][b].reshape(kd.shape), rtol=1e-5, atol=1e-6)
        tot += (1.0 * (reg + cls) + 0.005 * kd.sum()) * B; reg_t += reg * B; cls_t += cls * B; kd_t += kd.sum() * 0.005 * B
        for i in range(B):
            k = b * B + i
            np.testing.assert_array_equal(np.asarray(out["labels"][i], dtype=np.float32).reshape(-1, 5), g[f"label_img{k}"])
            for ti in range(3):
                np.testing.assert_array_equal(np.asarray(out["per_teacher"][ti][i], dtype=np.float32).reshape(-1, 6), g[f"teacher{ti}_img{k}"])
        preds = P.logits_to_ground_truth(out["logits_s"], S, 0.3, 0.5)
        for i in range(B):
            np.testing.assert_array_equal(np.asarray(preds[i], dtype=np.float32).reshape(-1, 6), g[f"pred_img{b * B + i}"])
    np.testing.assert_allclose(tot / N, g["val_loss"], rtol=1e-5)
    np.testing.assert_allclose(reg_t / N, g["scalar.Test/Regression_loss"], rtol=1e-5)
    np.testing.assert_allclose(cls_t / N, g["scalar.Test/Class_loss"], rtol=1e-5)
    np.testing.assert_allclose(kd_t / N, g["scalar.Test/KD"], rtol=1e-5)


# ---- BASELINE configs[4]'s architecture (D4) pinned on the reference's own classes: tools/oracle/make_golden.py golden_net_d4 / golden_step_d4
D4_MODS = [("rgb", 3, 41), ("thermal", 1, 42), ("audio", 8, 43)]


@pytest.mark.parametrize("mod,cin,seed", D4_MODS)
def test_net_d4_768_eval(golden_dir, mod, cin, seed):
    """`YetAnotherEfficientDet(compound_coef=4, in_channels=c)` in eval mode at 768^2, B = 1 (src/YetAnotherEfficientDet.py:608-629:
    fpn width 224, 7 cells, 4 head layers, backbone b4) - the oracle's D4 tables against the reference's, not against D2's."""
    g = np.load(os.path.join(golden_dir, f"net_d4_768_eval_{mod}.npz"))
    spec, st = make_state(4, cin, seed, mod)
    x = synth_inputs(1, 768, seed=44)[mod]
    with torch.no_grad():
        (c, r, a), f = O.forward(st, x, 4, False)
    assert c.shape == (1, 110484, 20) and [tuple(u.shape[1:]) for u in f] == [(224, 96, 96), (224, 48, 48), (224, 24, 24), (224, 12, 12), (224, 6, 6)]
    check_summary(g, "cls", c); check_summary(g, "reg", r); check_summary(g, "anchors", a, 1e-6, 1e-7)
    np.testing.assert_array_equal(a[0, ::997].numpy(), g["anchors.sample"])
    for i, u in enumerate(f):
        check_summary(g, f"feat{i}", u)


def test_net_d4_train_fwd_bwd(golden_dir):
    g = np.load(os.path.join(golden_dir, "net_d4_256_train_audio.npz"))
    spec, st = make_state(4, 8, 43, "audio")
    st = grad_state(st)
    x = synth_inputs(2, 256, seed=45)["audio"]
    assert [b.idx for b in spec.blocks if b.skip] == [int(b) for b in g["drop_blocks"]]
    masks = {int(b): torch.from_numpy(m) for b, m in zip(g["drop_blocks"], g["drop_masks"])}
    (c, r, a), f = O.forward(st, x, 4, True, masks)
    loss = c.sum() * 0.01 + (r ** 2).mean() + sum((u ** 2).mean() for u in f)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    check_summary(g, "cls", c); check_summary(g, "reg", r)
    gn = {}
    for k, v in st.items():
        if v.requires_grad and v.grad is not None:
            top = ".".join(k.split(".")[:2]) if k.startswith("bifpn") else k.split(".")[0]
            gn[top] = gn.get(top, 0.0) + float(v.grad.double().pow(2).sum())
    for k in g.files:
        if k.startswith("gradnorm."):
            assert abs(gn[k[9:]] ** 0.5 - float(g[k])) <= 2e-3 * float(g[k]), (k, gn[k[9:]] ** 0.5, float(g[k]))
        if k.startswith("grad.") and k.endswith(".head"):
            name = k[5:-5]
            check_summary(g, "grad." + name, st[name].grad, 2e-3, 1e-4)
        if k.startswith("stat.") and k.endswith(".head"):
            name = k[5:-5]
            check_summary(g, "stat." + name, st[name], 1e-5, 1e-6)
    assert int(st["backbone_net.model._bn0.num_batches_tracked"]) == int(g["nbt"])


D4_STEP_MODS = {"rgb": (3, 51), "depth": (3, 52), "thermal": (1, 53)}


def test_step_d4(golden_dir):
    """the reference's `ModelWithNMSLoss` built from D4 nets directly (256^2, B = 2): pseudo-label rows bit for bit, losses, gradients, Adam"""
    g = np.load(os.path.join(golden_dir, "step_d4_256_pairwise.npz"))
    S, B = 256, 2
    teachers = {k: make_state(4, cin, seed, k, cls_bias=-2.0)[1] for k, (cin, seed) in D4_STEP_MODS.items()}
    _, st = make_state(4, 8, 54, "audio")
    st = grad_state(st)
    batch = synth_inputs(B, S, seed=71)
    masks = {int(b): torch.from_numpy(m) for b, m in zip(g["drop_blocks"], g["drop_masks"])}
    out = ST.distill_forward(st, teachers, batch, S, 4, masks)
    for ti in range(3):
        for i in range(B):
            np.testing.assert_array_equal(out["per_teacher"][ti][i].reshape(-1, 6), g[f"teacher{ti}_img{i}"])
    np.testing.assert_allclose(out["reg"].detach().numpy(), g["reg"], rtol=1e-4)
    np.testing.assert_allclose(out["cls"].detach().numpy(), g["cls"], rtol=1e-4)
    np.testing.assert_allclose(torch.stack(out["kd"]).detach().numpy(), g["kd"], rtol=1e-5)
    loss = ST.total_loss(out)
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    loss.backward()
    params = {k: v for k, v in st.items() if v.requires_grad}
    grads = {k: v.grad for k, v in params.items() if v.grad is not None}
    for k in g.files:
        if k.startswith("grad.") and k.endswith(".head"):
            name = k[5:-5]
            check_summary(g, "grad." + name, grads[name], 2e-3, 1e-4)
    with torch.no_grad():
        ST.adam_step(params, grads, {})
    for k in g.files:
        if k.startswith("adam.") and k.endswith(".head"):
            name = k[5:-5]
            check_summary(g, "adam." + name, params[name], 1e-5, 1e-6)
