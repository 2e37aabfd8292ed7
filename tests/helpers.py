"""Shared test helpers: the state recipe used by tools/oracle/make_golden.py, repeated with the oracle."""
import os

import numpy as np
import torch

from mm_distillnet_amd.arch import make_spec
from mm_distillnet_amd.synth import synth_state, synth_inputs, calibrate_bn_
from oracle import effdet_ref as O

_CACHE = {}


def make_state(coef, cin, seed, calib_mod, cls_bias=-4.0):
    """hash weights + oracle BN calibration on a 4x256^2 batch (same as make_golden.make_state)."""
    key = (coef, cin, seed, calib_mod, cls_bias)
    if key not in _CACHE:
        spec = make_spec(coef, cin)
        st = synth_state(spec, seed=seed, cls_bias=cls_bias)

        def tf(state, x, mom):
            state["_bn_momentum"] = mom
            masks = {b.idx: torch.ones(x.shape[0]) for b in spec.blocks if b.skip}
            with torch.no_grad():
                O.forward(state, x, coef, True, masks)
            del state["_bn_momentum"]

        calibrate_bn_(st, tf, synth_inputs(4, 256, seed=1000 + seed)[calib_mod], seed=seed)
        _CACHE[key] = (spec, st)
    spec, st = _CACHE[key]
    return spec, {k: v.clone() for k, v in st.items()}


def check_summary(g, name, t, rtol=1e-4, atol=1e-5):
    """Compare tensor `t` with the head/sum/l2 summary stored under `name.*` in golden npz `g`."""
    t = t.detach().double().reshape(-1).cpu()
    assert int(g[name + ".numel"]) == t.numel(), name
    n = g[name + ".head"].shape[0]
    scale = max(float(g[name + ".absmax"]), 1e-30)
    np.testing.assert_allclose(t[:n].numpy(), g[name + ".head"].astype(np.float64), rtol=rtol, atol=atol * scale,
                               err_msg=name + ".head")
    l2 = float(g[name + ".l2"])
    assert abs(float(t.norm()) - l2) <= rtol * l2 + atol * scale, (name, float(t.norm()), l2)
    # sums of signed values cancel: tolerance relative to l2*sqrt(n)
    tol_sum = rtol * l2 * (t.numel() ** 0.5) + atol * scale
    assert abs(float(t.sum()) - float(g[name + ".sum"])) <= tol_sum, (name, float(t.sum()), float(g[name + ".sum"]))


def grad_state(st):
    return {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
            for k, v in st.items()}
