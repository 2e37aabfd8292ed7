"""ORACLE (test infrastructure, never shipped as product code).

CPU restatement, in plain PyTorch-CPU fp32 ops, of the reference EfficientDet used by
MM-DistillNet.  It is functional: `forward(state, x, ...)` over a flat state dict whose keys are the
reference's state-dict keys, so the same tensors drive the reference (in this container), this
oracle, and the HIP engine.  torch autograd differentiates it, which gives the backward reference.

Follows (reference file:line):
  TF-SAME conv / max-pool ......... src/YetAnotherEfficientNet.py:27-104
  swish ........................... src/YetAnotherEfficientNet.py:126-147
  drop-connect .................... src/YetAnotherEfficientNet.py:173-182
  MBConv block .................... src/YetAnotherEfficientNet.py:402-485
  backbone + taps ................. src/YetAnotherEfficientDet.py:535-572
  SeparableConvBlock .............. src/YetAnotherEfficientDet.py:154-192
  BiFPN fast attention ............ src/YetAnotherEfficientDet.py:320-392
  Regressor / Classifier .......... src/YetAnotherEfficientDet.py:445-532
  Anchors ......................... src/YetAnotherEfficientDet.py:71-151
  whole net ....................... src/YetAnotherEfficientDet.py:605-685

Parity pin: tests/golden/net_*.npz were produced by tools/oracle/make_golden.py from the
reference's own modules (imported in the build container) and this file is checked against them by
tests/test_oracle_golden.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module.
"""
from __future__ import annotations

import itertools
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
BN_MOM = 0.01

_COEF = {0: (1.0, 1.0), 1: (1.0, 1.1), 2: (1.1, 1.2), 3: (1.2, 1.4), 4: (1.4, 1.8), 5: (1.6, 2.2), 6: (1.8, 2.6)}
_STAGES = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
           (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
_FPN_W = [64, 88, 112, 160, 224, 288, 384, 384]
_FPN_N = [3, 4, 5, 6, 7, 7, 8, 8]
_HEAD_N = [3, 3, 3, 4, 4, 4, 5, 5]
_BB = [0, 1, 2, 3, 4, 5, 6, 6]
_ASCALE = [4., 4., 4., 4., 4., 4., 4., 5.]


def _rf(f, w):
    f = f * w
    n = max(8, int(f + 4) // 8 * 8)
    if n < 0.9 * f:
        n += 8
    return int(n)


def block_table(compound_coef: int):
    """[(kernel, stride, cin, cout, expand, se, skip)] for efficientnet-b{_BB[coef]}."""
    w, d = _COEF[_BB[compound_coef]]
    out = []
    for (r, k, s, e, i, o) in _STAGES:
        ci, co = _rf(i, w), _rf(o, w)
        for j in range(int(math.ceil(d * r))):
            cin = ci if j == 0 else co
            out.append((k, s if j == 0 else 1, cin, co, e, max(1, int(cin * 0.25)), j > 0 and cin == co))
    return out


def swish(x):
    return x * torch.sigmoid(x)


def same_pad_2d(x, k, s):
    h, w = x.shape[-2:]
    eh = (math.ceil(w / s) - 1) * s - w + k
    ev = (math.ceil(h / s) - 1) * s - h + k
    l, t = eh // 2, ev // 2
    return F.pad(x, [l, eh - l, t, ev - t])


# Oracle-only switch for the build's bf16 mixed-precision mode (BASELINE config 5; the reference itself has no reduced-
# precision path): when True, the two operands of every 1x1 convolution outside the squeeze-excite block are rounded to
# bf16 (round-to-nearest-even) and multiplied/accumulated in fp32 - the arithmetic of mmd_pwconv_fwd_bf16.  Straight-
# through in autograd (the backward stays fp32).
BF16_PW = False


def _bf16_ste(t):
    return t + (t.to(torch.bfloat16).to(t.dtype) - t).detach()


# Oracle-only tap: when a dict, forward() records the output of every unit the HIP engine materialises - "stem" (eval: activated),
# "blk<i>" (MBConv block outputs), "bifpn.<c>.<down-channel | conv*_up | conv*_down>" (NCHW tensors, detached) - so a test can hand each
# unit of the HIP net exactly the inputs this restatement saw ("teacher forcing": rounding differences cannot compound across units).
TAP = None


def _tap(name, x):
    if TAP is not None:
        TAP[name] = x.detach().clone()
    return x



def pw_conv(x, weight, bias=None):
    if BF16_PW:
        x, weight = _bf16_ste(x), _bf16_ste(weight)
    return F.conv2d(x, weight, bias)


def conv_same(x, weight, bias=None, stride=1, groups=1):
    k = weight.shape[-1]
    if k == 1 and groups == 1 and stride == 1:
        return pw_conv(x, weight, bias)
    return F.conv2d(same_pad_2d(x, k, stride), weight, bias, stride=stride, groups=groups)


def maxpool_same(x):
    # zero padding (NOT -inf) then 3x3 s2 max-pool
    return F.max_pool2d(same_pad_2d(x, 3, 2), 3, 2)


def batchnorm(state, prefix, x, training):
    w, b = state[prefix + ".weight"], state[prefix + ".bias"]
    rm, rv = state[prefix + ".running_mean"], state[prefix + ".running_var"]
    if training:
        # "_bn_momentum" is an oracle-only side channel used by BN calibration (momentum 1.0)
        y = F.batch_norm(x, rm, rv, w, b, True, state.get("_bn_momentum", BN_MOM), BN_EPS)
        nbt = prefix + ".num_batches_tracked"
        if nbt in state:
            state[nbt] += 1
        return y
    return F.batch_norm(x, rm, rv, w, b, False, BN_MOM, BN_EPS)


def mbconv(state, p, x, blk, training, drop_rate, drop_mask):
    k, s, cin, cout, e, se, skip = blk
    inp = x
    if e != 1:
        x = conv_same(x, state[p + "._expand_conv.conv.weight"])
        x = swish(batchnorm(state, p + "._bn0", x, training))
    x = conv_same(x, state[p + "._depthwise_conv.conv.weight"], stride=s, groups=x.shape[1])
    x = swish(batchnorm(state, p + "._bn1", x, training))
    sq = F.adaptive_avg_pool2d(x, 1)
    sq = swish(F.conv2d(sq, state[p + "._se_reduce.conv.weight"], state[p + "._se_reduce.conv.bias"]))
    sq = F.conv2d(sq, state[p + "._se_expand.conv.weight"], state[p + "._se_expand.conv.bias"])
    x = torch.sigmoid(sq) * x
    x = conv_same(x, state[p + "._project_conv.conv.weight"])
    x = batchnorm(state, p + "._bn2", x, training)
    if skip:
        if training and drop_rate:
            keep = 1.0 - drop_rate
            # drop_mask: [B] tensor of 0/1 (floor(keep + U)); reference draws it with torch.rand
            x = x / keep * drop_mask.view(-1, 1, 1, 1).to(x.dtype)
        x = x + inp
    return x


def backbone(state, x, coef, training, drop_masks):
    p = "backbone_net.model"
    x = conv_same(x, state[p + "._conv_stem.conv.weight"], stride=2)
    x = swish(batchnorm(state, p + "._bn0", x, training))
    if not training:
        _tap("stem", x)
    blocks = block_table(coef)
    n = len(blocks)
    fm, last = [], None
    for i, blk in enumerate(blocks):
        rate = 0.2 * float(i) / n
        dm = None if drop_masks is None else drop_masks.get(i)
        if training and blk[6] and rate and dm is None:
            raise ValueError("training-mode oracle needs an explicit drop mask for block %d" % i)
        x = _tap(f"blk{i}", mbconv(state, f"{p}._blocks.{i}", x, blk, training, rate, dm))
        if blk[1] == 2:
            fm.append(last)
        elif i == n - 1:
            fm.append(x)
        last = x
    return fm[2:]  # p3, p4, p5


def sepconv(state, p, x, training, norm=True, act=False, bn_prefix=None):
    x = conv_same(x, state[p + ".depthwise_conv.conv.weight"], groups=x.shape[1])
    x = pw_conv(x, state[p + ".pointwise_conv.conv.weight"], state[p + ".pointwise_conv.conv.bias"])
    if norm:
        x = batchnorm(state, bn_prefix or (p + ".bn"), x, training)
    if act:
        x = swish(x)
    return x


def _fw(state, key):
    w = F.relu(state[key])
    return w / (torch.sum(w, dim=0) + 1e-4)


def _up(x):
    return F.interpolate(x, scale_factor=2, mode="nearest")


def bifpn_cell(state, p, feats, first, training):
    def dc(name, x):
        x = pw_conv(x, state[f"{p}.{name}.0.conv.weight"], state[f"{p}.{name}.0.conv.bias"])
        return _tap(f"{p}.{name}", batchnorm(state, f"{p}.{name}.1", x, training))

    if first:
        p3, p4, p5 = feats
        p6_in = maxpool_same(dc("p5_to_p6", p5))
        p7_in = maxpool_same(p6_in)
        p3_in, p4_in, p5_in = dc("p3_down_channel", p3), dc("p4_down_channel", p4), dc("p5_down_channel", p5)
    else:
        p3_in, p4_in, p5_in, p6_in, p7_in = feats
    sc = lambda n, x: _tap(f"{p}.{n}", sepconv(state, f"{p}.{n}", x, training))
    w = _fw(state, p + ".p6_w1"); p6_up = sc("conv6_up", swish(w[0] * p6_in + w[1] * _up(p7_in)))
    w = _fw(state, p + ".p5_w1"); p5_up = sc("conv5_up", swish(w[0] * p5_in + w[1] * _up(p6_up)))
    w = _fw(state, p + ".p4_w1"); p4_up = sc("conv4_up", swish(w[0] * p4_in + w[1] * _up(p5_up)))
    w = _fw(state, p + ".p3_w1"); p3_out = sc("conv3_up", swish(w[0] * p3_in + w[1] * _up(p4_up)))
    if first:
        p4_in, p5_in = dc("p4_down_channel_2", p4), dc("p5_down_channel_2", p5)
    w = _fw(state, p + ".p4_w2")
    p4_out = sc("conv4_down", swish(w[0] * p4_in + w[1] * p4_up + w[2] * maxpool_same(p3_out)))
    w = _fw(state, p + ".p5_w2")
    p5_out = sc("conv5_down", swish(w[0] * p5_in + w[1] * p5_up + w[2] * maxpool_same(p4_out)))
    w = _fw(state, p + ".p6_w2")
    p6_out = sc("conv6_down", swish(w[0] * p6_in + w[1] * p6_up + w[2] * maxpool_same(p5_out)))
    w = _fw(state, p + ".p7_w2")
    p7_out = sc("conv7_down", swish(w[0] * p7_in + w[1] * maxpool_same(p6_out)))
    return p3_out, p4_out, p5_out, p6_out, p7_out


def head(state, p, feats, layers, out_per_anchor, training, sigmoid):
    outs = []
    for lvl, f in enumerate(feats):
        for i in range(layers):
            f = sepconv(state, f"{p}.conv_list.{i}", f, training, norm=True, act=True,
                        bn_prefix=f"{p}.bn_list.{lvl}.{i}")
        f = sepconv(state, f"{p}.header", f, training, norm=False)
        f = f.permute(0, 2, 3, 1).contiguous()
        outs.append(f.view(f.shape[0], -1, out_per_anchor))
    o = torch.cat(outs, dim=1)
    return torch.sigmoid(o) if sigmoid else o


def anchors_for(image_size: int, coef: int = 2) -> torch.Tensor:
    """[1, A, 4] (y1,x1,y2,x2) fp32; float64 math then cast (YetAnotherEfficientDet.py:116-147)."""
    scales = np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)])
    ratios = [(1.0, 1.0), (1.4, 0.7), (0.7, 1.4)]
    allb = []
    for stride in [8, 16, 32, 64, 128]:
        lv = []
        for scale, ratio in itertools.product(scales, ratios):
            base = _ASCALE[coef] * stride * scale
            ax, ay = base * ratio[0] / 2.0, base * ratio[1] / 2.0
            x = np.arange(stride / 2, image_size, stride)
            xv, yv = np.meshgrid(x, x)
            xv, yv = xv.reshape(-1), yv.reshape(-1)
            b = np.vstack((yv - ay, xv - ax, yv + ay, xv + ax)).swapaxes(0, 1)
            lv.append(np.expand_dims(b, 1))
        allb.append(np.concatenate(lv, axis=1).reshape(-1, 4))
    return torch.from_numpy(np.vstack(allb).astype(np.float32)).unsqueeze(0)


def forward(state: Dict[str, torch.Tensor], x: torch.Tensor, coef: int = 2, training: bool = False,
            drop_masks: Optional[Dict[int, torch.Tensor]] = None, num_classes: int = 20):
    """Returns ([classification, regression, anchors], (p3..p7)) like the reference (:662-685)."""
    feats = backbone(state, x, coef, training, drop_masks)
    for c in range(_FPN_N[coef]):
        feats = bifpn_cell(state, f"bifpn.{c}", feats, c == 0, training)
    reg = head(state, "regressor", feats, _HEAD_N[coef], 4, training, False)
    cls = head(state, "classifier", feats, _HEAD_N[coef], num_classes, training, True)
    return [cls, reg, anchors_for(x.shape[-1], coef)], feats
