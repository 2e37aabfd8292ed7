"""Drop-in Python surface of the reference for the hot path: same class names, constructor
arguments, forward signatures / return structures and state-dict keys, backed by the HIP engine.

  YetAnotherEfficientDet ...... src/YetAnotherEfficientDet.py:605-685
  MTALoss ..................... src/loss/MTALoss.py:9-77
  YetAnotherFocalLoss ......... src/loss/YetAnotherFocalLoss.py:23-190
  load_model, extract_criterions_from_config ... src/utils/utils.py:441-590,1556-1668

The nn.Parameters of `YetAnotherEfficientDet` are zero-copy strided VIEWS into the engine's flat
parameter buffer (kernel-native layouts), their `.grad`s come back as views of the flat gradient buffer,
so `torch.optim.*`, `state_dict()/load_state_dict()` and `loss.backward()` work unchanged while every
FLOP runs in the HIP kernels.  The whole network is ONE autograd node.  (The fast training path is
step.DistillEngine, which also fuses the losses, pseudo-labels and optimizer and is graph-captured.)
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .arch import make_spec
from .engine import Net, Feat
from .layout import state_layout

call = _lib.call


class _Node(nn.Module):
    pass


def _strided_view(flat: torch.Tensor, e) -> torch.Tensor:
    """torch-shaped view of a native-layout entry of the flat buffer."""
    shape = e.shape
    if e.kind == "dw":
        C, _, k, _ = shape
        return flat.as_strided((C, 1, k, k), (1, k * k * C, k * C, C), e.off)
    if e.kind == "stem":
        co, ci = shape[0], shape[1]
        return flat.as_strided((co, ci, 3, 3), (e.native[1], 9, 3, 1), e.off)
    if e.kind == "se_w" and e.native[0] != shape[0]:
        C, S = shape[0], shape[1]
        return flat.as_strided((C, S, 1, 1), (1, C, 1, 1), e.off)
    return flat[e.off:e.off + e.n].view(shape)


class _DetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        net: Net = module._net
        train = module.training
        B = x.shape[0]
        net.begin_step()
        ds = None
        if train:
            keep = module._keep
            ds = torch.floor(keep + torch.rand(keep.shape[0], B, device=x.device)) / keep
        cls, reg, feats = net.forward(x.contiguous(), train=train, drop_scale=ds)
        ctx.module = module
        ctx.train = train
        ctx.feat_shapes = [(f.B, f.H, f.W, f.C) for f in feats]
        # outputs are cloned out of the bump arena (the next forward of this net reuses those addresses)
        outs = [cls.clone(), reg.clone()] + [f.z.view(f.B, f.H, f.W, f.C).permute(0, 3, 1, 2).clone() for f in feats]
        ctx.save_for_backward(outs[0])
        return tuple(outs)

    @staticmethod
    def backward(ctx, dcls, dreg, *dfeats):
        module = ctx.module
        net: Net = module._net
        if not ctx.train:
            raise RuntimeError("backward through an eval-mode forward is not supported by the HIP engine")
        (cls,) = ctx.saved_tensors
        dlogit = torch.empty_like(cls)
        dcls = torch.zeros_like(cls) if dcls is None else dcls.contiguous()
        call("mmd_sigmoid_bwd", dcls, cls, dlogit, cls.numel())
        dreg = torch.zeros_like(cls[..., :4]).contiguous() if dreg is None else dreg.contiguous()
        dfe = []
        for d, (B, H, W, C) in zip(dfeats, ctx.feat_shapes):
            dfe.append(None if d is None else d.permute(0, 2, 3, 1).contiguous().view(B * H * W, C))
        net.ps.grad.zero_()
        net.backward(dlogit, dreg, dfe)
        grads = [g.clone() for g in module._grad_views]
        return (None, None, *grads)


class YetAnotherEfficientDet(nn.Module):
    def __init__(self, num_classes=20, compound_coef=2, load_weights=False, input_data_config=None,
                 output_data_config=None, in_channels=3, features_from='efficientnet', integration_mode=None,
                 device=None, **kwargs):
        super().__init__()
        if features_from not in ['efficientnet', 'header']:
            raise NotImplementedError()
        if features_from == 'header':
            raise NotImplementedError("features_from='header' is outside the hot path (shipped cfg uses 'efficientnet')")
        self.compound_coef = compound_coef
        self.features_from = features_from
        self.num_classes = num_classes
        self.spec = make_spec(compound_coef, in_channels, num_classes)
        dev = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        self._bind(Net(self.spec, dev, trainable=True))
        from .synth import synth_state
        self.load_state_dict(synth_state(self.spec, seed=0, cls_bias=-4.0))   # deterministic init; real weights via load_state_dict

    # ---- parameter tree -------------------------------------------------------------
    def _bind(self, net: Net):
        object.__setattr__(self, "_net", net)
        ps = net.ps
        for name in list(self._modules.keys()):
            del self._modules[name]
        self._grad_views: List[torch.Tensor] = []
        self._param_list: List[nn.Parameter] = []
        bn_index = {n: i for i, n in enumerate(ps.bn_names)}
        for key, shape, kind in state_layout(self.spec):
            parts = key.split(".")
            node = self
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _Node())
                node = node._modules[p]
            leaf = parts[-1]
            if key in ps.entries:
                e = ps.entries[key]
                par = nn.Parameter(_strided_view(ps.flat, e))
                node.register_parameter(leaf, par)
                self._param_list.append(par)
                self._grad_views.append(_strided_view(ps.grad, e))
                continue
            bname = key.rsplit(".", 1)[0]
            o, c = ps.bn_off[bname], ps.bn_c[bname]
            if kind in ("bn_w", "bn_b"):
                base = ps.gamma_off if kind == "bn_w" else ps.beta_off
                par = nn.Parameter(ps.flat[base + o:base + o + c])
                node.register_parameter(leaf, par)
                self._param_list.append(par)
                self._grad_views.append(ps.grad[base + o:base + o + c])
            elif kind == "bn_rm":
                node.register_buffer(leaf, ps.rmean[o:o + c])
            elif kind == "bn_rv":
                node.register_buffer(leaf, ps.rvar[o:o + c])
            else:
                node.register_buffer(leaf, ps.nbt[bn_index[bname]])
        keep = [1.0 - b.drop_rate for b in self.spec.blocks if b.skip]
        object.__setattr__(self, "_keep", torch.tensor(keep, device=ps.flat.device).view(-1, 1))

    def _apply(self, fn, recurse=True):
        probe = fn(torch.empty(0, device=self._net.ps.flat.device))
        if probe.device != self._net.ps.flat.device:
            state = self._net.ps.export_state()
            net = Net(self.spec, probe.device, trainable=True)
            net.load_state(state)
            self._bind(net)
        return self

    def load_state_dict(self, state_dict, strict=True):
        res = super().load_state_dict(state_dict, strict=strict)
        self._net.refresh()
        return res

    def train(self, mode: bool = True):
        if not mode and self.training:
            self._net.refresh()           # eval uses the folded running statistics
        return super().train(mode)

    def freeze_bn(self):
        raise NotImplementedError("freeze_bn is not used on the hot path")

    def forward(self, inputs: torch.Tensor):
        net = self._net
        if not self.training:
            net.refresh()
        outs = _DetFn.apply(self, inputs, *self._param_list)
        cls, reg = outs[0], outs[1]
        anchors = net.anchors(inputs.shape[-1]).unsqueeze(0)
        return [cls, reg, anchors], tuple(outs[2:])


# ---------------------------------------------------------------------------------------- losses
class _MTAFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, T, p, nlev, nteach, *maps):
        fs = maps[:nlev]
        dev = fs[0].device
        losses = torch.zeros(nlev, device=dev)
        das, rows = [], []
        for l, f in enumerate(fs):
            B, C, H, W = f.shape
            fr = f.permute(0, 2, 3, 1).contiguous().view(B * H * W, C)
            a_s = torch.empty(B * H * W, device=dev)
            call("mmd_mta_attention", fr, a_s, B * H * W, C, float(p))
            ats = []
            for k in range(nteach):
                t = maps[nlev + k * nlev + l]
                tr = t.permute(0, 2, 3, 1).contiguous().view(B * H * W, C)
                a_t = torch.empty(B * H * W, device=dev)
                call("mmd_mta_attention", tr, a_t, B * H * W, C, float(p))
                ats.append(a_t)
            da = torch.empty(B * H * W, device=dev)
            call("mmd_mta_kl", a_s, ats[0], ats[1] if nteach > 1 else None, ats[2] if nteach > 2 else None, nteach, B,
                 H * W, float(T), losses[l:l + 1], da, 1.0, 0)
            das.append(da); rows.append(fr)
        ctx.p = p; ctx.nlev = nlev; ctx.nteach = nteach
        ctx.shapes = [tuple(f.shape) for f in fs]
        ctx.save_for_backward(*das, *rows)
        return losses

    @staticmethod
    def backward(ctx, g):
        n = ctx.nlev
        das, rows = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        out = []
        for l in range(n):
            B, C, H, W = ctx.shapes[l]
            df = torch.empty(B * H * W, C, device=g.device)
            call("mmd_mta_attention_bwd", rows[l], (das[l] * g[l]).contiguous(), df, B * H * W, C, float(ctx.p), 0)
            out.append(df.view(B, H, W, C).permute(0, 3, 1, 2))
        return (None, None, None, None, *out, *([None] * (n * ctx.nteach)))


class MTALoss(nn.Module):
    def __init__(self, T=9.0, p=2.0):
        super().__init__()
        self.p = float(p)
        self.T = float(T)

    def forward(self, g_s, g_t):
        if torch.is_tensor(g_t[0]):
            teachers = [list(g_t)]
        else:
            teachers = [list(t) for t in g_t]
        if len(teachers) > 3:
            raise Exception("MTALoss supports up to 3 teachers on the HIP path")
        flat = [m for t in teachers for m in t]
        return _MTAFn.apply(self.T, self.p, len(g_s), len(teachers), *g_s, *[m.detach() for m in flat])


class _FocalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls, reg, anchors, boxes, nbox, maxg):
        B, A, NC = cls.shape
        dev = cls.device
        assign = torch.empty(B * A, dtype=torch.int32, device=dev)
        npos = torch.empty(B, dtype=torch.int32, device=dev)
        acc = torch.empty(2 * B, dtype=torch.float64, device=dev)
        out = torch.zeros(2, device=dev)
        dcls = torch.empty_like(cls); dreg = torch.empty_like(reg)
        call("mmd_focal_loss", cls.contiguous(), reg.contiguous(), anchors, boxes, nbox, maxg, B, A, NC, assign, npos, acc,
             out, dcls, dreg, 1.0, 0, None)
        ctx.save_for_backward(dcls, dreg)
        return out[0:1].clone(), out[1:2].clone()

    @staticmethod
    def backward(ctx, g_reg, g_cls):
        dcls, dreg = ctx.saved_tensors
        return dcls * g_cls, dreg * g_reg, None, None, None, None


class YetAnotherFocalLoss(nn.Module):
    def forward(self, prediction, annotations, **kwargs):
        classifications, regressions, anchors = prediction
        dev = classifications.device
        B = classifications.shape[0]
        maxg = max(1, max((int(np.shape(a)[0]) if np.ndim(a) > 0 else 0) for a in annotations))
        boxes = torch.full((B, maxg, 5), -1.0)
        nbox = torch.zeros(B, dtype=torch.int32)
        for i, a in enumerate(annotations):
            if np.size(a):
                a = np.asarray(a, dtype=np.float32).reshape(-1, 5)
                boxes[i, :a.shape[0]] = torch.from_numpy(a)
                nbox[i] = a.shape[0]
        return _FocalFn.apply(classifications, regressions, anchors[0].contiguous(), boxes.to(dev), nbox.to(dev), maxg)


# ---------------------------------------------------------------------------------------- factories
_MODALITY_CHANNELS = {'rgb': 3, 'audio_static': 8, 'audio_student': 8, 'depth': 3, 'thermal': 1, None: 3}
_MODALITY_PATH = {'rgb': "trained_models/yet-another-efficientdet-d2-rgb.pth",
                  'audio_static': "trained_models/yet-another-efficientdet-d2-audio.pth",
                  'depth': "trained_models/yet-another-efficientdet-d2-depth.pth",
                  'thermal': "trained_models/yet-another-efficientdet-d2-thermal.pth"}


def filter_state_dict(model_keys: Dict[str, torch.Size], pretrained: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Key remapping of src/utils/utils.py:327-411 for the plain detector: strip `module.`, map the generator's
    `model_backbones.<mod>` / `model_necks.<mod>` names onto `backbone_net` / `bifpn`, keep shape-matching entries."""
    out = {}
    remap = [("model_backbones.%s" % m, "backbone_net") for m in ("audio", "thermal", "depth", "rgb")] + \
            [("model_necks.%s" % m, "bifpn") for m in ("audio", "thermal", "depth", "rgb")] + \
            [("model_regressor", "regressor"), ("model_classifier", "classifier")]
    for k, v in pretrained.items():
        cands = [k, k[7:] if k.startswith("module.") else k]
        for src, dst in remap:
            cands += [c.replace(src, dst) for c in list(cands) if src in c]
        for c in cands:
            if c in model_keys and tuple(v.shape) == tuple(model_keys[c]):
                out[c] = v
    return out


def load_model(model_type: str, config, modality: Optional[str] = None, device=None) -> YetAnotherEfficientDet:
    """Model factory with the reference's names (src/utils/utils.py:441-590).  compound_coef defaults to 2 like the
    reference; the cfg key `compound_coef` (absent upstream) selects another EfficientDet size."""
    if 'YetAnotherEfficientDet' not in model_type or 'Generator' in model_type:
        raise Exception(f"Unsupported model type {model_type} provided")
    if modality not in _MODALITY_CHANNELS:
        raise Exception(f"Unsupported modality={modality} on load model")
    coef = int(config.get('compound_coef', 2)) if hasattr(config, 'get') else 2
    model = YetAnotherEfficientDet(compound_coef=coef, in_channels=_MODALITY_CHANNELS[modality],
                                   features_from=config['features_from'] if 'features_from' in config else 'efficientnet',
                                   device=device)
    path = _MODALITY_PATH.get(modality, "trained_models/yet-another-efficientdet-d2.pth")
    paths = [path] + (["trained_models/yet-another-efficientdet-d2-embedding.pth"] if 'embedding' in model_type else [])
    for pth in paths:
        if os.path.exists(pth):
            sd = model.state_dict()
            sd.update(filter_state_dict({k: v.shape for k, v in sd.items()}, torch.load(pth, map_location="cpu")))
            model.load_state_dict(sd)
    return model


def extract_criterions_from_config(config):
    """(criterion_main, criterion_div, criterion_kd) for the shipped recipe (src/utils/utils.py:1556-1668)."""
    if config['main_loss'] != 'YetAnotherFocalLoss':
        raise Exception(f"Unsupported main_loss {config['main_loss']} on the HIP path")
    kd = None
    if config.get('kd_loss', 'None') == 'MTALoss':
        kd = MTALoss(T=config['T'], p=config['p'])
    elif config.get('kd_loss', 'None') not in ('None', None):
        raise Exception(f"Unsupported kd_loss {config['kd_loss']} on the HIP path")
    return YetAnotherFocalLoss(), None, kd
