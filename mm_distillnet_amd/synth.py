"""Deterministic synthetic state dicts (there is no network: `trained_models/*.pth` are unavailable).

Every tensor is generated from a hash of its state-dict key and a seed, so the reference (imported in
the build container to make golden vectors), the oracle and the HIP engine all see the same weights
without shipping 32 MB files.  Gains are chosen so eval-mode activations stay
O(1) through the 23 MBConv blocks and 5 BiFPN cells, BN running statistics are non-trivial, and
the classifier header bias is about -4 so that a realistic handful of anchors pass conf 0.3.
"""
from __future__ import annotations

import os
import sys
import zlib
from typing import Dict

import numpy as np
import torch

from .arch import NetSpec
from .layout import state_layout


def _rng(key: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([zlib.crc32(key.encode()), seed & 0xFFFFFFFF]))


def synth_state(spec: NetSpec, seed: int = 0, cls_bias: float = -4.0) -> Dict[str, torch.Tensor]:
    out: Dict[str, torch.Tensor] = {}
    for key, shape, kind in state_layout(spec):
        g = _rng(key, seed)
        if kind == "pw":
            fan_in = shape[1]
            # inputs to expand convs are BN outputs (unit-ish second moment); inputs to project
            # convs are swish*gate (second moment ~0.1); heads/BiFPN see dw outputs.
            gain = 1.0
            a = g.standard_normal(shape, dtype=np.float32) * np.float32(gain / np.sqrt(fan_in))
        elif kind == "dw":
            k = shape[-1]
            a = g.standard_normal(shape, dtype=np.float32) * np.float32(1.0 / k)
        elif kind == "stem":
            fan_in = shape[1] * 9
            a = g.standard_normal(shape, dtype=np.float32) * np.float32(1.0 / np.sqrt(fan_in))
        elif kind == "se_w":
            a = g.standard_normal(shape, dtype=np.float32) * np.float32(1.0 / np.sqrt(shape[1]))
        elif kind == "bias":
            a = g.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
            if key == "classifier.header.pointwise_conv.conv.bias":
                a = a + np.float32(cls_bias)
        elif kind == "bn_w":
            a = g.uniform(0.8, 1.2, shape).astype(np.float32)
        elif kind == "bn_b":
            a = g.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
        elif kind == "bn_rm":
            a = np.zeros(shape, dtype=np.float32)
        elif kind == "bn_rv":
            a = np.ones(shape, dtype=np.float32)
        elif kind == "bn_nbt":
            out[key] = torch.zeros((), dtype=torch.int64)
            continue
        elif kind == "fuse":
            # mostly positive (relu keeps them), one entry may be <= 0 to exercise the relu gate
            a = g.uniform(0.3, 1.5, shape).astype(np.float32)
            if zlib.crc32(key.encode()) % 7 == 0:
                a[-1] = np.float32(-0.25)
        else:
            raise KeyError(kind)
        out[key] = torch.from_numpy(np.ascontiguousarray(a))
    return out


def calibrate_bn_(state: Dict[str, torch.Tensor], train_forward, x: torch.Tensor, seed: int = 0) -> None:
    """Make BN running statistics consistent with the activations (as in a trained net).

    `train_forward(state, x, bn_momentum)` must run ONE train-mode forward that overwrites
    running_mean/var with the batch statistics (momentum 1.0, all drop-connect masks = 1); either the
    oracle (tests) or the HIP engine (bench) can provide it.  Afterwards the stats are perturbed by a
    key-hashed factor so eval-mode BN is not an exact normaliser, and num_batches_tracked is reset.
    """
    train_forward(state, x, 1.0)
    for key in list(state.keys()):
        if key.endswith(".running_var"):
            g = _rng(key, seed)
            v = state[key]
            f = torch.from_numpy(g.uniform(0.8, 1.25, tuple(v.shape)).astype(np.float32)).to(v.device)
            v.copy_(torch.clamp(v, min=0.05) * f)
            m = state[key.replace("running_var", "running_mean")]
            d = torch.from_numpy((g.standard_normal(tuple(v.shape)) * 0.1).astype(np.float32)).to(v.device)
            m.add_(d * torch.sqrt(v))
        elif key.endswith(".num_batches_tracked"):
            state[key].zero_()


def synth_inputs(batch: int, image_size: int, seed: int = 24, audio_raw: int = 128) -> Dict[str, torch.Tensor]:
    """Synthetic modality batches per SURVEY.md §8(d): rgb ~N(0,1) (ImageNet-normalised), thermal and
    depth ~U[0,1] (÷255 only), audio = 8-channel dB mel ~N(-40,15) at 128x128 bicubic-resized to SxS
    (the build's counterpart of `Resizer`, src/datasets/transformations.py:442-448)."""
    g = torch.Generator().manual_seed(seed)
    rgb = torch.randn(batch, 3, image_size, image_size, generator=g)
    thermal = torch.rand(batch, 1, image_size, image_size, generator=g)
    depth = torch.rand(batch, 3, image_size, image_size, generator=g)
    raw = torch.randn(batch, 8, audio_raw, audio_raw, generator=g) * 15.0 - 40.0
    audio = torch.nn.functional.interpolate(raw, size=(image_size, image_size), mode="bicubic", align_corners=False)
    return {"rgb": rgb, "thermal": thermal, "depth": depth, "audio": audio.contiguous()}


# ---- synthetic stand-ins with a realistic workload (bench.py; train.py with cfg synthetic_teacher_candidates) ----
def calibrated_state(spec, seed, x, device, cls_bias=-4.0):
    """hash weights + BN running stats calibrated with one train-mode pass of the HIP engine."""
    st = synth_state(spec, seed=seed, cls_bias=cls_bias)
    from .engine import Net
    net = Net(spec, device, trainable=True)

    def tf(state, xin, mom):
        net.load_state(state)
        net.bn_momentum = mom
        net.begin_step()
        net.forward(xin.to(device), train=True, drop_scale=None)
        torch.cuda.synchronize()
        ex = net.ps.export_state()
        for k in state:
            if "running_" in k:
                state[k].copy_(ex[k])

    calibrate_bn_(st, tf, x, seed=seed)
    del net
    torch.cuda.empty_cache()
    return st


def tune_teacher_bias(spec, state, x, device, target_per_image=40):
    """Shift the classifier header bias uniformly so that ~target candidates per image pass (score > 0.3, class car):
    random-weight teachers otherwise emit either nothing or thousands of boxes, which is not the workload."""
    import math
    from .engine import Net
    net = Net(spec, device, trainable=False)
    xd = x.to(device)
    for it in range(8):         # sigmoid saturates: iterate until the count is in range
        net.load_state(state)
        net.begin_step()
        logit, _, _ = net.forward(xd, train=False, raw_logits=True)      # pre-sigmoid: probabilities saturate in fp32
        logit = logit.double()
        best, arg = logit.max(2)
        car = best[arg == 6]
        n_now = int((car > math.log(0.3 / 0.7)).sum().item())
        tgt = target_per_image * logit.shape[0]
        if os.environ.get("MMD_BENCH_DEBUG"):
            print("tune it %d: over-threshold car candidates %d (target %d), car anchors %d" % (it, n_now, tgt, car.numel()), file=sys.stderr)
        if 0.5 * tgt <= n_now <= 1.5 * tgt or car.numel() <= tgt:
            break
        v = torch.sort(car, descending=True)[0][tgt].item()
        state["classifier.header.pointwise_conv.conv.bias"] += float(math.log(0.3 / 0.7) - v)
    del net
    torch.cuda.empty_cache()
