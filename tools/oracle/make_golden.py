#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules (build container only).

    python tools/oracle/make_golden.py            # writes tests/golden/

The reference (read-only at /root/reference) is imported through tools/oracle/refshim.py.  Every
fixture stores inputs that cannot be regenerated from a seed plus the reference's outputs
(full tensors when tiny, otherwise head-slices + sums + L2 norms).  The GPU box never sees the
reference; it only sees these vectors.  tests/test_oracle_golden.py checks oracle/ against them
and the `-m gpu` tests check the HIP engine against both.
"""
import configparser
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import refshim  # noqa: E402

refshim.install()
torch.set_num_threads(8)

from mm_distillnet_amd.arch import make_spec  # noqa: E402
from mm_distillnet_amd.layout import state_layout  # noqa: E402
from mm_distillnet_amd.synth import synth_state, synth_inputs, calibrate_bn_  # noqa: E402
from oracle import effdet_ref as O  # noqa: E402

from src.YetAnotherEfficientDet import YetAnotherEfficientDet  # noqa: E402
import src.YetAnotherEfficientNet as REN  # noqa: E402
from src.loss.MTALoss import MTALoss  # noqa: E402
from src.loss.YetAnotherFocalLoss import YetAnotherFocalLoss  # noqa: E402
import src.utils.utils as RU  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def summ(t: torch.Tensor, n=64):
    t = t.detach().double().reshape(-1)
    return {"head": t[:n].float().numpy(), "sum": np.float64(t.sum().item()),
            "l2": np.float64(t.norm().item()), "absmax": np.float64(t.abs().max().item() if t.numel() else 0.0),
            "numel": np.int64(t.numel())}


def put(d, name, t, n=64):
    for k, v in summ(t, n).items():
        d[f"{name}.{k}"] = v


def make_state(coef, cin, seed, calib_mod, cls_bias=-4.0):
    """Shared recipe (tests repeat it): hash weights + oracle BN calibration on a 4x256^2 batch."""
    spec = make_spec(coef, cin)
    st = synth_state(spec, seed=seed, cls_bias=cls_bias)

    def tf(state, x, mom):
        state["_bn_momentum"] = mom
        masks = {b.idx: torch.ones(x.shape[0]) for b in spec.blocks if b.skip}
        with torch.no_grad():
            O.forward(state, x, coef, True, masks)
        del state["_bn_momentum"]

    calibrate_bn_(st, tf, synth_inputs(4, 256, seed=1000 + seed)[calib_mod], seed=seed)
    return spec, st


class MaskedDropConnect:
    """Replaces the reference's RNG-driven drop_connect with a recorded, replayable mask."""

    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.calls = []

    def __call__(self, inputs, p, training):
        if not training:
            return inputs
        keep = 1 - p
        mk = torch.floor(keep + torch.rand(inputs.shape[0], generator=self.g))
        self.calls.append(mk)
        return inputs / keep * mk.view(-1, 1, 1, 1)


def ref_model(coef, cin, st):
    m = YetAnotherEfficientDet(compound_coef=coef, in_channels=cin)
    m.load_state_dict(st)
    return m


def cfg(image_size, student="YetAnotherEfficientDet_D2"):
    c = configparser.ConfigParser()
    c["DEFAULT"] = {"student": student, "image_size": str(image_size), "conf_threshold": "0.3",
                    "nms_threshold": "0.5", "valid_labels": "car"}
    return c["DEFAULT"]


VALID = {"labels_txt2i": {"car": 6}, "labels_i2txt": {6: "car"},
         "predictions_txt2i": {"car": 6}, "predictions_i2txt": {6: "car"}}


def golden_state_keys():
    for coef in (2, 4):
        m = YetAnotherEfficientDet(compound_coef=coef, in_channels=8)
        rows = [[k, list(v.shape)] for k, v in m.state_dict().items()]
        json.dump(rows, open(os.path.join(OUT, f"state_keys_d{coef}_c8.json"), "w"))
    print("state keys written")


def golden_net():
    S, B = 128, 2
    for mod, cin, seed in [("rgb", 3, 11), ("thermal", 1, 12), ("audio", 8, 13)]:
        spec, st = make_state(2, cin, seed, mod)
        x = synth_inputs(B, S, seed=24)[mod]
        m = ref_model(2, cin, st).eval()
        d = {}
        with torch.no_grad():
            (c, r, a), f = m(x)
        put(d, "cls", c); put(d, "reg", r); put(d, "anchors", a)
        for i, u in enumerate(f):
            put(d, f"feat{i}", u)
        np.savez(os.path.join(OUT, f"net_d2_eval_{mod}.npz"), **d)
        print("net eval", mod, d["cls.sum"], d["reg.l2"])
    # anchors at 512 (full tensor is 786 KB -> keep checksums + a strided sample)
    m = YetAnotherEfficientDet(compound_coef=2, in_channels=3)
    a = m.anchors(torch.zeros(1, 3, 512, 512))
    d = {}
    put(d, "anchors512", a)
    d["anchors512.sample"] = a[0, ::997].numpy()
    np.savez(os.path.join(OUT, "anchors_d2_512.npz"), **d)

    # training mode (student): fwd + bwd of a scalar, running-stat update, injected drop masks
    spec, st = make_state(2, 8, 13, "audio")
    x = synth_inputs(B, S, seed=25)["audio"]
    dc = MaskedDropConnect(7)
    REN.drop_connect = dc
    m = ref_model(2, 8, st).train()
    (c, r, a), f = m(x)
    loss = c.sum() * 0.01 + (r ** 2).mean() + sum((u ** 2).mean() for u in f)
    loss.backward()
    d = {"loss": np.float64(loss.item())}
    put(d, "cls", c); put(d, "reg", r)
    for i, u in enumerate(f):
        put(d, f"feat{i}", u)
    d["drop_masks"] = torch.stack(dc.calls).numpy()
    d["drop_blocks"] = np.array([b.idx for b in spec.blocks if b.skip], dtype=np.int64)
    sd = m.state_dict()
    gn = {}
    for k, p in m.named_parameters():
        top = ".".join(k.split(".")[:2]) if k.startswith("bifpn") else k.split(".")[0]
        gn[top] = gn.get(top, 0.0) + float(p.grad.double().pow(2).sum())
    for k, v in gn.items():
        d[f"gradnorm.{k}"] = np.float64(v ** 0.5)
    for k in ["backbone_net.model._conv_stem.conv.weight", "backbone_net.model._blocks.5._depthwise_conv.conv.weight",
              "backbone_net.model._blocks.9._se_reduce.conv.weight", "backbone_net.model._blocks.22._bn2.weight",
              "bifpn.0.p4_w2", "bifpn.2.conv5_down.depthwise_conv.conv.weight", "bifpn.0.p5_to_p6.0.conv.weight",
              "regressor.header.pointwise_conv.conv.bias", "classifier.conv_list.1.pointwise_conv.conv.weight",
              "classifier.bn_list.3.2.weight"]:
        put(d, "grad." + k, dict(m.named_parameters())[k].grad)
    for k in ["backbone_net.model._bn0.running_mean", "backbone_net.model._blocks.10._bn1.running_var",
              "bifpn.3.conv4_up.bn.running_var", "regressor.bn_list.4.0.running_mean"]:
        put(d, "stat." + k, sd[k])
    d["nbt"] = np.int64(sd["backbone_net.model._bn0.num_batches_tracked"].item())
    np.savez(os.path.join(OUT, "net_d2_train_audio.npz"), **d)
    print("net train", d["loss"])


def golden_losses():
    torch.manual_seed(3)
    S = 128
    anchors = YetAnotherEfficientDet(compound_coef=2, in_channels=3).anchors(torch.zeros(1, 3, S, S))
    A = anchors.shape[1]
    crit = YetAnotherFocalLoss()
    cases = {
        "mixed": [np.array([[10, 12, 60, 70, 6], [64, 30, 120, 100, 6], [5, 5, 9, 8, 6]], dtype=np.float32),
                  np.zeros((0,), dtype=np.float32),
                  np.array([[0, 0, 128, 128, 6]], dtype=np.float32)],
        "all_empty": [np.zeros((0,), dtype=np.float32), [], np.zeros((0,), dtype=np.float32)],
        "ignore_band": [np.array([[20, 20, 52, 52, 6], [21, 23, 50, 58, 3]], dtype=np.float32),
                        np.array([[40, 8, 100, 40, 0]], dtype=np.float32),
                        np.array([[100, 100, 101, 100, 6]], dtype=np.float32)],
    }
    for name, ann in cases.items():
        B = len(ann)
        g = torch.Generator().manual_seed(len(name))
        cls = torch.sigmoid(torch.randn(B, A, 20, generator=g) * 2 - 2)
        cls[0, :50] = 0.0          # exercise the clamp (zero gradient region)
        cls[1, :50] = 1.0
        reg = torch.randn(B, A, 4, generator=g) * 0.3
        cls.requires_grad_(True); reg.requires_grad_(True)
        rl, cl = crit([cls, reg, anchors], ann)
        d = {"seed": np.int64(len(name)), "image_size": np.int64(S),
             "reg_loss": rl.detach().numpy(), "cls_loss": cl.detach().numpy()}
        put(d, "cls_in", cls); put(d, "reg_in", reg)
        for i, a in enumerate(ann):
            d[f"ann{i}"] = np.asarray(a, dtype=np.float32).reshape(-1, 5)
        if rl.requires_grad or cl.requires_grad:
            (rl.sum() * 1.0 + cl.sum() * 1.0).backward()
            put(d, "dcls", cls.grad); put(d, "dreg", reg.grad)
            d["dcls.sample"] = cls.grad.reshape(-1)[::211].numpy().copy()
            d["dreg.sample"] = reg.grad.reshape(-1)[::53].numpy().copy()
        np.savez_compressed(os.path.join(OUT, f"loss_focal_{name}.npz"), **d)
        print("focal", name, rl.detach().numpy(), cl.detach().numpy())

    # MTA: pairwise and list mode, stock (T=9,p=2) and a peaky variant (T=0.05) whose gradient is not ~0
    for name, T, scale in [("stock", 9.0, 1.0), ("peaky", 0.05, 3.0)]:
        g = torch.Generator().manual_seed(17)
        sizes = [16, 8, 4, 2, 1]
        fs = [(torch.randn(2, 12, s, s, generator=g) * scale).requires_grad_(True) for s in sizes]
        fts = [[torch.randn(2, 12, s, s, generator=g) * scale for s in sizes] for _ in range(3)]
        crit = MTALoss(T=T, p=2)
        d = {"T": np.float64(T)}
        for i, f in enumerate(fs):
            d[f"fs{i}"] = f.detach().numpy()
        for k, ft in enumerate(fts):
            for i, f in enumerate(ft):
                d[f"ft{k}_{i}"] = f.numpy()
        pair = crit(fs, fts[0])
        pair.sum().backward()
        d["pair"] = pair.detach().numpy()
        for i, f in enumerate(fs):
            d[f"pair_dfs{i}"] = f.grad.numpy().copy(); f.grad = None
        lst = crit(fs, fts)
        lst.sum().backward()
        d["list"] = lst.detach().numpy()
        for i, f in enumerate(fs):
            d[f"list_dfs{i}"] = f.grad.numpy().copy(); f.grad = None
        np.savez_compressed(os.path.join(OUT, f"loss_mta_{name}.npz"), **d)
        print("mta", name, d["pair"], d["list"])


def golden_postproc():
    S, B = 128, 3
    anchors = YetAnotherEfficientDet(compound_coef=2, in_channels=3).anchors(torch.zeros(1, 3, S, S))
    A = anchors.shape[1]
    g = torch.Generator().manual_seed(5)
    cls = torch.sigmoid(torch.randn(B, A, 20, generator=g) * 1.5 - 3.0)
    # make class 6 win often so the class filter keeps a decent number
    cls[:, :, 6] = torch.sigmoid(torch.randn(B, A, generator=g) * 2.0 - 1.5)
    cls[2] = cls[2] * 0.2          # image with no candidate at all
    reg = torch.randn(B, A, 4, generator=g) * 0.4
    gts = RU.logits_to_ground_truth([cls, reg, anchors], None, VALID, cfg(S), include_scores=True)
    d = {"seed": np.int64(5), "image_size": np.int64(S)}
    put(d, "cls_in", cls); put(d, "reg_in", reg)
    for i, gt in enumerate(gts):
        d[f"gt{i}"] = np.asarray(gt, dtype=np.float32).reshape(-1, 6)
        print("postproc img", i, d[f"gt{i}"].shape)
    np.savez_compressed(os.path.join(OUT, "postproc_d2_128.npz"), **d)


def golden_step():
    tm = refshim.load_train_methods()
    S, B = 256, 2
    mods = {"rgb": (3, 21), "depth": (3, 22), "thermal": (1, 23)}
    bias = {"rgb": -2.0, "depth": -3.2, "thermal": -2.0}     # so that every teacher contributes boxes
    states = {k: make_state(2, cin, seed, k, cls_bias=bias[k])[1] for k, (cin, seed) in mods.items()}
    spec, st_s = make_state(2, 8, 24, "audio")
    batch = synth_inputs(B, S, seed=31)
    only = os.environ.get("GOLDEN_STEP_VARIANTS", "pairwise,list,augmented,rgb1,listaug").split(",")
    for variant, cls_name in [("pairwise", "ModelWithNMSLoss"), ("list", "ModelWithNMSKDListLoss"),
                              ("augmented", "ModelWithNMSLossAugmented"), ("rgb1", "ModelWithNMSLoss"),
                              ("listaug", "ModelWithNMSKDListLossAugmented")]:
        if variant not in only:
            continue
        # "rgb1" = BASELINE configs[1]: use_thermal = use_depth = False -> the ModuleDict holds the RGB teacher only
        # (train.py:123-135; the step module loops over whichever teachers are present, train_methods.py:441)
        tmods = ("rgb",) if variant == "rgb1" else ("rgb", "depth", "thermal")
        teachers = torch.nn.ModuleDict()
        for k in tmods:        # reference insertion order (train.py:123-135)
            teachers[k] = ref_model(2, mods[k][0], states[k])
        teachers.eval()
        for p_ in teachers.parameters():
            p_.requires_grad = False
        dc = MaskedDropConnect(9)
        REN.drop_connect = dc
        student = ref_model(2, 8, st_s).train()
        model = getattr(tm, cls_name)(student, teachers, YetAnotherFocalLoss(), None, MTALoss(T="9", p="2"),
                                       cfg(S), VALID)
        opt = torch.optim.Adam(student.parameters(), lr=1e-4, betas=(0.9, 0.999))
        opt.zero_grad()
        if variant == "listaug":        # `label` = RGB frames of other recordings (train_set.yield_batch), augment=True: a 4th list entry
            aug_rgb = synth_inputs(B, S, seed=57)["rgb"]
            res = model(batch["rgb"], batch["thermal"], batch["depth"], batch["audio"], aug_rgb, augment=True)
        elif variant == "augmented":      # augment=True: audio merge, teacher feature averaging, labels of image 0 -> image 1
            res = model(batch["rgb"], batch["thermal"], batch["depth"], batch["audio"].clone(), None, augment=True)
        else:
            res = model(batch["rgb"], batch["thermal"], batch["depth"], batch["audio"], None)
        regl, clsl, kdl = res[0], res[1], res[2]
        loss_main = torch.mean(torch.stack(regl)) + torch.mean(torch.stack(clsl))
        loss = 1.0 * loss_main + 0.005 * torch.sum(torch.stack(kdl))
        loss.backward()
        d = {"reg": regl[0].detach().numpy(), "cls": clsl[0].detach().numpy(),
             "kd": torch.stack(kdl).detach().numpy(), "loss": np.float64(loss.item()),
             "drop_masks": torch.stack(dc.calls).numpy(),
             "drop_blocks": np.array([b.idx for b in spec.blocks if b.skip], dtype=np.int64)}
        gn = {}
        for k, p_ in student.named_parameters():
            if p_.grad is None:
                continue
            top = ".".join(k.split(".")[:2]) if k.startswith("bifpn") else k.split(".")[0]
            gn[top] = gn.get(top, 0.0) + float(p_.grad.double().pow(2).sum())
        for k, v in gn.items():
            d[f"gradnorm.{k}"] = np.float64(v ** 0.5)
        watch = ["backbone_net.model._conv_stem.conv.weight", "backbone_net.model._blocks.12._project_conv.conv.weight",
                 "bifpn.4.conv3_up.pointwise_conv.conv.weight", "classifier.header.pointwise_conv.conv.bias",
                 "regressor.header.pointwise_conv.conv.weight", "bifpn.1.p5_w2"]
        named = dict(student.named_parameters())
        for k in watch:
            if named[k].grad is None:
                print("no grad for", k)
                continue
            put(d, "grad." + k, named[k].grad)
        opt.step()
        for k in watch:
            put(d, "adam." + k, named[k])
        # merged labels as seen by the loss: recompute through the reference helpers for the record
        with torch.no_grad():
            per = []
            for k in tmods:
                pred, _ = teachers[k](batch[k])
                per.append(RU.logits_to_ground_truth(pred, None, VALID, cfg(S), include_scores=True))
        if variant == "listaug":
            with torch.no_grad():
                pred, _ = teachers["rgb"](aug_rgb)
                per.append(RU.logits_to_ground_truth(pred, None, VALID, cfg(S), include_scores=True))
        for ti, lab in enumerate(per):
            for i in range(B):
                d[f"teacher{ti}_img{i}"] = np.asarray(lab[i], dtype=np.float32).reshape(-1, 6)
        np.savez_compressed(os.path.join(OUT, f"step_d2_256_{variant}.npz"), **d)
        print("step", variant, d["reg"], d["cls"], d["kd"].reshape(-1)[:5], d["loss"],
              [d[f"teacher{t}_img{i}"].shape[0] for t in range(len(per)) for i in range(B)])


D4_MODS = [("rgb", 3, 41), ("thermal", 1, 42), ("audio", 8, 43)]


def golden_net_d4():
    """BASELINE configs[4]'s architecture pinned on the reference's own classes ("D4@768 B = 1"): the reference builds
    `YetAnotherEfficientDet(compound_coef=4, in_channels=c)` from the tables at src/YetAnotherEfficientDet.py:608-629 (fpn width 224,
    7 BiFPN cells, 4 head layers, backbone b4); only `load_model` hard-codes D2, so the nets are constructed directly.
    (1) eval forward at 768^2, B = 1, in_channels 3 / 1 / 8; (2) train-mode forward + backward at 256^2, B = 2, with recorded
    drop-connect masks, running-stat update included."""
    S = 768
    for mod, cin, seed in D4_MODS:
        spec, st = make_state(4, cin, seed, mod)
        x = synth_inputs(1, S, seed=44)[mod]
        m = ref_model(4, cin, st).eval()
        d = {}
        with torch.no_grad():
            (c, r, a), f = m(x)
        put(d, "cls", c); put(d, "reg", r); put(d, "anchors", a)
        d["anchors.sample"] = a[0, ::997].numpy()
        for i, u in enumerate(f):
            put(d, f"feat{i}", u)
        np.savez(os.path.join(OUT, f"net_d4_768_eval_{mod}.npz"), **d)
        print("net d4 eval", mod, tuple(c.shape), d["cls.sum"], d["reg.l2"], [tuple(u.shape) for u in f])
    S, B = 256, 2
    spec, st = make_state(4, 8, 43, "audio")
    x = synth_inputs(B, S, seed=45)["audio"]
    dc = MaskedDropConnect(17)
    REN.drop_connect = dc
    m = ref_model(4, 8, st).train()
    (c, r, a), f = m(x)
    loss = c.sum() * 0.01 + (r ** 2).mean() + sum((u ** 2).mean() for u in f)
    loss.backward()
    d = {"loss": np.float64(loss.item())}
    put(d, "cls", c); put(d, "reg", r)
    for i, u in enumerate(f):
        put(d, f"feat{i}", u)
    d["drop_masks"] = torch.stack(dc.calls).numpy()
    d["drop_blocks"] = np.array([b.idx for b in spec.blocks if b.skip], dtype=np.int64)
    assert d["drop_masks"].shape[0] == d["drop_blocks"].shape[0], (d["drop_masks"].shape, d["drop_blocks"].shape)
    sd = m.state_dict()
    named = dict(m.named_parameters())
    gn = {}
    for k, p in named.items():
        top = ".".join(k.split(".")[:2]) if k.startswith("bifpn") else k.split(".")[0]
        gn[top] = gn.get(top, 0.0) + float(p.grad.double().pow(2).sum())
    for k, v in gn.items():
        d[f"gradnorm.{k}"] = np.float64(v ** 0.5)
    for k in ["backbone_net.model._conv_stem.conv.weight", "backbone_net.model._blocks.7._depthwise_conv.conv.weight",
              "backbone_net.model._blocks.13._se_reduce.conv.weight", "backbone_net.model._blocks.31._bn2.weight",
              "backbone_net.model._blocks.30._expand_conv.conv.weight",
              "bifpn.0.p4_w2", "bifpn.6.conv5_down.depthwise_conv.conv.weight", "bifpn.0.p5_to_p6.0.conv.weight",
              "bifpn.3.conv6_up.pointwise_conv.conv.weight",
              "regressor.header.pointwise_conv.conv.bias", "classifier.conv_list.3.pointwise_conv.conv.weight",
              "classifier.bn_list.4.3.weight"]:
        put(d, "grad." + k, named[k].grad)
    for k in ["backbone_net.model._bn0.running_mean", "backbone_net.model._blocks.20._bn1.running_var",
              "bifpn.5.conv4_up.bn.running_var", "regressor.bn_list.4.3.running_mean"]:
        put(d, "stat." + k, sd[k])
    d["nbt"] = np.int64(sd["backbone_net.model._bn0.num_batches_tracked"].item())
    np.savez(os.path.join(OUT, "net_d4_256_train_audio.npz"), **d)
    print("net d4 train", d["loss"], {k: round(float(v), 5) for k, v in d.items() if k.startswith("gradnorm.")})


def golden_step_d4():
    """A whole `ModelWithNMSLoss` step (src/optimization/train_methods.py:425-560) on D4 nets built directly - three frozen D4 teachers
    and the 8-channel D4 student at 256^2, B = 2: losses, gradients, one Adam step, the teachers' pseudo-label rows."""
    tm = refshim.load_train_methods()
    S, B = 256, 2
    mods = {"rgb": (3, 51), "depth": (3, 52), "thermal": (1, 53)}
    bias = {"rgb": -2.0, "depth": -2.0, "thermal": -2.0}
    states = {k: make_state(4, cin, seed, k, cls_bias=bias[k])[1] for k, (cin, seed) in mods.items()}
    spec, st_s = make_state(4, 8, 54, "audio")
    batch = synth_inputs(B, S, seed=71)
    teachers = torch.nn.ModuleDict()
    for k in ("rgb", "depth", "thermal"):
        teachers[k] = ref_model(4, mods[k][0], states[k])
    teachers.eval()
    for p_ in teachers.parameters():
        p_.requires_grad = False
    dc = MaskedDropConnect(19)
    REN.drop_connect = dc
    student = ref_model(4, 8, st_s).train()
    model = tm.ModelWithNMSLoss(student, teachers, YetAnotherFocalLoss(), None, MTALoss(T="9", p="2"),
                                cfg(S, "YetAnotherEfficientDet_D4"), VALID)
    opt = torch.optim.Adam(student.parameters(), lr=1e-4, betas=(0.9, 0.999))
    opt.zero_grad()
    res = model(batch["rgb"], batch["thermal"], batch["depth"], batch["audio"], None)
    regl, clsl, kdl = res[0], res[1], res[2]
    loss_main = torch.mean(torch.stack(regl)) + torch.mean(torch.stack(clsl))
    loss = 1.0 * loss_main + 0.005 * torch.sum(torch.stack(kdl))
    loss.backward()
    d = {"reg": regl[0].detach().numpy(), "cls": clsl[0].detach().numpy(),
         "kd": torch.stack(kdl).detach().numpy(), "loss": np.float64(loss.item()),
         "drop_masks": torch.stack(dc.calls).numpy(),
         "drop_blocks": np.array([b.idx for b in spec.blocks if b.skip], dtype=np.int64)}
    gn = {}
    named = dict(student.named_parameters())
    for k, p_ in named.items():
        if p_.grad is None:
            continue
        top = ".".join(k.split(".")[:2]) if k.startswith("bifpn") else k.split(".")[0]
        gn[top] = gn.get(top, 0.0) + float(p_.grad.double().pow(2).sum())
    for k, v in gn.items():
        d[f"gradnorm.{k}"] = np.float64(v ** 0.5)
    watch = ["backbone_net.model._conv_stem.conv.weight", "backbone_net.model._blocks.16._project_conv.conv.weight",
             "bifpn.6.conv3_up.pointwise_conv.conv.weight", "classifier.header.pointwise_conv.conv.bias",
             "regressor.header.pointwise_conv.conv.weight", "bifpn.1.p5_w2", "regressor.conv_list.3.depthwise_conv.conv.weight"]
    for k in watch:
        if named[k].grad is None:
            print("no grad for", k)
            continue
        put(d, "grad." + k, named[k].grad)
    opt.step()
    for k in watch:
        put(d, "adam." + k, named[k])
    with torch.no_grad():
        per = []
        for k in ("rgb", "depth", "thermal"):
            pred, _ = teachers[k](batch[k])
            per.append(RU.logits_to_ground_truth(pred, None, VALID, cfg(S, "YetAnotherEfficientDet_D4"), include_scores=True))
    for ti, lab in enumerate(per):
        for i in range(B):
            d[f"teacher{ti}_img{i}"] = np.asarray(lab[i], dtype=np.float32).reshape(-1, 6)
    np.savez_compressed(os.path.join(OUT, "step_d4_256_pairwise.npz"), **d)
    print("step d4", d["reg"], d["cls"], d["kd"].reshape(-1)[:5], d["loss"],
          [d[f"teacher{t}_img{i}"].shape[0] for t in range(3) for i in range(B)])


def synth_detections(seed=41, n_batches=5, bs=4, S=512):
    """Synthetic evaluation set: per image pseudo ground truth [m,5] (x1,y1,x2,y2,label) and student detections [n,6]
    (x1,y1,x2,y2,score,label) - jittered copies of some ground-truth boxes, false positives, wrong-class boxes, images with no
    detection / no ground truth.  Shared by the golden generator and tests/test_model_cpu.py (deterministic from the seed)."""
    rng = np.random.RandomState(seed)
    all_pred, all_lab = [], []
    for b in range(n_batches):
        bp, bl = [], []
        for i in range(bs):
            m = int(rng.randint(0, 6)) if (b, i) != (0, 1) else 0
            gt = []
            for _ in range(m):
                x1, y1 = rng.randint(0, S - 80, size=2)
                w, h = rng.randint(20, 160, size=2)
                gt.append([x1, y1, min(x1 + w, S), min(y1 + h, S), int(rng.choice([6, 6, 6, 14, 1]))])
            gt = np.array(gt, dtype=np.float32).reshape(-1, 5)
            det = []
            for g_ in gt:
                r = rng.rand()
                if r < 0.75:      # detected, jittered (some land below IoU 0.75 / 0.5)
                    j = rng.randint(-14, 15, size=4) * (1.0 if rng.rand() < 0.6 else 2.5)
                    lab = g_[4] if rng.rand() < 0.9 else 14.0
                    det.append([max(g_[0] + j[0], 0), max(g_[1] + j[1], 0), min(g_[2] + j[2], S), min(g_[3] + j[3], S),
                                0.3 + 0.7 * rng.rand(), lab])
                if r > 0.9:       # duplicate detection of the same object
                    det.append([g_[0] + 2, g_[1] + 1, g_[2] - 3, g_[3] + 2, 0.3 + 0.5 * rng.rand(), g_[4]])
            for _ in range(int(rng.randint(0, 3))):      # false positives
                x1, y1 = rng.randint(0, S - 60, size=2)
                det.append([x1, y1, x1 + rng.randint(10, 60), y1 + rng.randint(10, 60), 0.3 + 0.4 * rng.rand(), 6.0])
            if (b, i) == (1, 2):
                det = []
            det = np.array(det, dtype=np.float32).reshape(-1, 6)
            det = np.floor(det * [1, 1, 1, 1, 1e4, 1]) / [1, 1, 1, 1, 1e4, 1]
            if det.shape[0]:
                det = det[np.argsort(-det[:, 4], kind="stable")]      # NMS emits rows in score order
            bp.append(det.astype(np.float32)); bl.append(gt)
        all_pred.append(bp); all_lab.append(bl)
    labels = [float(r[4]) for bl in all_lab for g_ in bl for r in g_]
    return all_pred, all_lab, labels


def golden_metrics():
    """The reference's evaluate() AP / CD table (src/utils/utils.py:2018-2181) and its building blocks
    get_batch_statistics (:1058-1136), ap_per_class (:1188-1252), compute_ap (:1255-1280), get_batch_central_distances
    (:993-1055) on a synthetic detection set.  evaluate() itself is driven with get_predictions_multiteacher patched to return
    that set (the models are never called) and its results.<rank>.csv read back."""
    import tempfile
    import pandas as pd
    S = 512
    all_pred, all_lab, labels = synth_detections(S=S)
    d = {"image_size": np.int64(S), "seed": np.int64(41)}
    flat_p = [p_ for bp in all_pred for p_ in bp]
    flat_l = [l_ for bl in all_lab for l_ in bl]
    d["pred_rows"] = np.concatenate(flat_p, 0); d["pred_counts"] = np.array([p_.shape[0] for p_ in flat_p], dtype=np.int64)
    d["lab_rows"] = np.concatenate(flat_l, 0); d["lab_counts"] = np.array([l_.shape[0] for l_ in flat_l], dtype=np.int64)
    for iou in (0.5, 0.75, 0.9):
        sm = []
        for bp, bl in zip(all_pred, all_lab):
            sm += RU.get_batch_statistics(bp, bl, iou)
        tp, sc, lb = [np.concatenate(x, 0) for x in zip(*sm)]
        d[f"tp@{iou}"] = tp; d[f"score@{iou}"] = sc; d[f"label@{iou}"] = lb
        p_, r_, ap_, f1_, cls_, score_ = RU.ap_per_class(tp, sc, lb, labels)
        d[f"precision@{iou}"] = p_; d[f"recall@{iou}"] = r_; d[f"ap@{iou}"] = ap_; d[f"f1@{iou}"] = f1_
        d[f"ap_class@{iou}"] = cls_; d[f"score_ratio@{iou}"] = np.float64(score_)
    cdx, cdy = [], []
    for bp, bl in zip(all_pred, all_lab):
        x_, y_ = RU.get_batch_central_distances(bp, bl, S, S)
        cdx.extend(x_); cdy.extend(y_)
    d["cd_x"] = np.array(cdx, dtype=np.float64); d["cd_y"] = np.array(cdy, dtype=np.float64)
    # evaluate(): the whole table
    tmp = tempfile.mkdtemp()
    c = configparser.ConfigParser()
    c["DEFAULT"] = {"exp_name": tmp, "rank": "0", "use_rgb": "True", "use_thermal": "True", "use_depth": "True",
                    "student": "YetAnotherEfficientDet_D2", "image_size": str(S)}

    class _Set:
        classes = [f"c{i}" for i in range(21)]

    def ragged(batch):      # evaluate()'s debug line calls np.array(batch).shape, which numpy >= 1.24 refuses for ragged lists
        o = np.empty(len(batch), dtype=object)
        for i_, a_ in enumerate(batch):
            o[i_] = a_
        return o

    saved = RU.get_predictions_multiteacher
    RU.get_predictions_multiteacher = lambda *a, **k: ([ragged(b_) for b_ in all_pred], [ragged(b_) for b_ in all_lab], labels)
    try:
        RU.evaluate({"rgb": None, "depth": None, "thermal": None}, torch.nn.Linear(2, 2), _Set(), c["DEFAULT"])
    finally:
        RU.get_predictions_multiteacher = saved
    row = pd.read_csv(os.path.join(tmp, "results.0.csv")).iloc[0]
    for k in ("AP@Ave", "AP@0.5", "AP@0.75", "CDx", "CDy"):
        d["table." + k] = np.float64(row[k])
    assert row["modality"] == "ALL"
    np.savez_compressed(os.path.join(OUT, "metrics_eval.npz"), **d)
    print("metrics", {k: float(d["table." + k]) for k in ("AP@Ave", "AP@0.5", "AP@0.75", "CDx", "CDy")},
          "n_pred", int(d["pred_counts"].sum()), "n_gt", int(d["lab_counts"].sum()))


VAL_S, VAL_N, VAL_B = 256, 4, 2          # validation golden: 4 samples of 256^2 in batches of 2


def val_states():
    """Teacher / student states of the validation golden (tests repeat it through helpers.make_state): the step goldens' teachers and a
    student whose classifier bias lets the eval-mode student emit detections."""
    mods = {"rgb": (3, 21), "depth": (3, 22), "thermal": (1, 23)}
    bias = {"rgb": -2.0, "depth": -3.2, "thermal": -2.0}
    tstates = {k: make_state(2, cin, seed, k, cls_bias=bias[k])[1] for k, (cin, seed) in mods.items()}
    spec, st_s = make_state(2, 8, 24, "audio", cls_bias=-2.0)
    return mods, tstates, spec, st_s


def golden_validate():
    """The reference's validate() (src/optimization/train_methods.py:1083-1185) and get_predictions_multiteacher
    (src/utils/utils.py:1720-1893) run as they are on a 4-sample synthetic set: eval-mode student, validate=True, per-batch loss sums
    times the sample count over len(val_set), the Test/* scalars; student detections + merged multi-teacher pseudo ground truth."""
    import inspect
    import tempfile
    tm = refshim.load_train_methods()
    S, N, B = VAL_S, VAL_N, VAL_B
    mods, tstates, spec, st_s = val_states()
    data = synth_inputs(N, S, seed=61)
    tmp = tempfile.mkdtemp()
    for i in range(N):                       # get_predictions_multiteacher drops a <timestamp>.all.txt next to each sample
        os.makedirs(os.path.join(tmp, f"drive{i}"))

    class Set(torch.utils.data.Dataset):
        valid_classes_dict = VALID
        data_path = tmp

        def __len__(self):
            return N

        def __getitem__(self, i):
            return data["rgb"][i], data["thermal"][i], data["depth"][i], data["audio"][i], [], f"drive{i}/{1000 + i}"

    teachers = torch.nn.ModuleDict()
    for k in ("rgb", "depth", "thermal"):
        teachers[k] = ref_model(2, mods[k][0], tstates[k])
    teachers.eval()
    student = ref_model(2, 8, st_s).train()          # validate() itself switches the student to eval()
    model = tm.ModelWithNMSLoss(student, teachers, YetAnotherFocalLoss(), None, MTALoss(T="9", p="2"), cfg(S), VALID)

    class Wrap(torch.nn.Module):                      # validate() reaches the student through model.module (DataParallel / DDP)
        def __init__(self, m):
            super().__init__()
            self.module = m

        def forward(self, *a, **k):
            return self.module(*a, **k)

    c = configparser.ConfigParser()
    c["DEFAULT"] = {"student": "YetAnotherEfficientDet_D2", "image_size": str(S), "conf_threshold": "0.3", "nms_threshold": "0.5",
                    "valid_labels": "car", "batch_size": str(B), "num_workers": "0", "use_thermal": "True", "use_depth": "True",
                    "use_rgb": "True", "w_main": "1.0", "w_div": "0.0", "w_kd": "0.005", "num_epoches": "1", "student_modality": "audio",
                    "use_labels": "False"}
    conf = c["DEFAULT"]

    class Writer:
        def __init__(self):
            self.rows = {}

        def add_scalar(self, tag, v, step):
            self.rows[tag] = float(v)

    # per-batch terms, recorded by wrapping the step module's forward
    per_batch = []
    inner = model.forward

    def spy(*a, **k):
        out = inner(*a, **k)
        per_batch.append((float(torch.sum(torch.stack(out[0]))), float(torch.sum(torch.stack(out[1]))),
                          torch.stack(out[2]).detach().numpy().copy(), bool(k.get("validate", False)), bool(k.get("augment", False))))
        return out

    model.forward = spy
    w = Writer()
    val_loss = tm.validate(Set(), Wrap(model), 0, conf, w)
    assert not student.training and all(pb[3] and not pb[4] for pb in per_batch)
    d = {"val_loss": np.float64(float(val_loss)), "n": np.int64(N), "batch": np.int64(B), "image_size": np.int64(S)}
    for tag, v in w.rows.items():
        d["scalar." + tag] = np.float64(v)
    d["batch_reg"] = np.array([pb[0] for pb in per_batch]); d["batch_cls"] = np.array([pb[1] for pb in per_batch])
    d["batch_kd"] = np.stack([pb[2] for pb in per_batch])
    # get_predictions_multiteacher, with the numpy >= 1.25 fix applied to its text in memory (`ndarray == []`), as for train_methods
    src = inspect.getsource(RU.get_predictions_multiteacher).replace("batch_labels[i] == []", "isListEmpty(batch_labels[i])")
    ns = dict(RU.__dict__)
    exec(compile(src, "get_predictions_multiteacher(patched)", "exec"), ns)
    preds, labs, flat = ns["get_predictions_multiteacher"](teachers, student, Set(), conf)
    k = 0
    for bp, bl in zip(preds, labs):
        for i in range(len(bl)):
            d[f"pred_img{k}"] = np.asarray(bp[i] if len(bp) else [], dtype=np.float32).reshape(-1, 6)
            d[f"label_img{k}"] = np.asarray(bl[i], dtype=np.float32).reshape(-1, 5)
            k += 1
    assert k == N
    d["labels_flat"] = np.asarray(flat, dtype=np.float64)
    # the teachers' own rows per image (what the tight GPU test feeds back in place of the GPU teachers' decode + NMS)
    with torch.no_grad():
        for ti, m in enumerate(("rgb", "depth", "thermal")):
            for b0 in range(0, N, B):          # in the loader's batches of B (conv results can differ in the last bit with the batch size)
                pred, _ = teachers[m](data[m][b0:b0 + B])
                rows = RU.logits_to_ground_truth(pred, None, VALID, cfg(S), include_scores=True)
                for i in range(B):
                    d[f"teacher{ti}_img{b0 + i}"] = np.asarray(rows[i], dtype=np.float32).reshape(-1, 6)
    np.savez_compressed(os.path.join(OUT, "validate_d2_256.npz"), **d)
    print("validate", d["val_loss"], w.rows, "batch reg/cls", d["batch_reg"], d["batch_cls"],
          "preds", [d[f"pred_img{i}"].shape[0] for i in range(N)], "labels", [d[f"label_img{i}"].shape[0] for i in range(N)])


if __name__ == "__main__":
    which = sys.argv[1:] or ["keys", "net", "losses", "postproc", "step", "metrics", "validate", "net_d4", "step_d4"]
    if "metrics" in which:
        golden_metrics()
    if "validate" in which:
        golden_validate()
    if "keys" in which:
        golden_state_keys()
    if "net" in which:
        golden_net()
    if "losses" in which:
        golden_losses()
    if "postproc" in which:
        golden_postproc()
    if "step" in which:
        golden_step()
    if "net_d4" in which:
        golden_net_d4()
    if "step_d4" in which:
        golden_step_d4()
