"""Device-resident parameter store and bump arenas for one detector.

MI355X-first memory layout: every trainable parameter of a net lives in ONE flat fp32 buffer in
kernel-native layout (1x1 weights [Cout,Cin] as in torch; depthwise weights tap-major [k*k,C]; the stem
weight [32,Kp] zero-padded so the im2col GEMM reads float4s), BatchNorm gammas / betas / running
stats are each contiguous over all 144 layers (one launch folds or updates all of them), gradients and
Adam moments mirror the parameter buffer (one coalesced optimizer pass, contiguous all-reduce
buckets).  State-dict import/export converts to/from the reference's torch layouts and key names
(src/utils/utils.py:327-411 key conventions).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from .arch import NetSpec, BN_EPS
from .layout import state_layout


def _r4(n: int) -> int:
    return (n + 3) // 4 * 4


@dataclass
class Entry:
    key: str
    kind: str
    shape: Tuple[int, ...]      # torch (reference) shape
    off: int                    # offset in the flat buffer (floats)
    n: int                      # native element count (may include padding)
    native: Tuple[int, ...]     # native 2-D/1-D shape


class ParamStore:
    def __init__(self, spec: NetSpec, device, with_grads: bool, stem_slot: int = 0):
        """stem_slot: floats reserved for the stem weight (>= its own size): nets that differ only in their input channels then share one
        layout for everything behind the stem (grouped frozen nets: engine.py pack mode)."""
        self.spec = spec
        self.device = device
        rows = state_layout(spec)
        self.entries: Dict[str, Entry] = {}
        self.order: List[str] = []
        off = 0
        conv_kinds = ("pw", "dw", "stem", "se_w", "bias", "fuse")
        # execution-ordered conv/bias/theta weights first: backbone -> bifpn -> regressor -> classifier
        def exec_rank(k: str) -> int:
            return 0 if k.startswith("backbone_net") else 1 if k.startswith("bifpn") else 2 if k.startswith("regressor") else 3
        ordered = sorted([r for r in rows if r[2] in conv_kinds], key=lambda r: exec_rank(r[0]))
        self.stem_kp = _r4(spec.in_channels * 9)
        for key, shape, kind in ordered:
            if kind == "se_w" and key.endswith("_se_expand.conv.weight"):
                native = (shape[1], shape[0])          # [S][C]: the gate kernel reads it coalesced over channels
            elif kind == "pw" or kind == "se_w":
                native = (shape[0], shape[1])
            elif kind == "dw":
                native = (shape[2] * shape[3], shape[0])
            elif kind == "stem":
                native = (shape[0], self.stem_kp)
            else:
                native = tuple(shape)
            n = 1
            for d in native:
                n *= d
            self.entries[key] = Entry(key, kind, tuple(shape), off, n, native)
            self.order.append(key)
            off += _r4(max(n, stem_slot) if kind == "stem" else n)
        self.n_conv = off
        # BN: gammas, then betas, contiguous; channel offsets shared with the running-stat buffers
        self.bn_names: List[str] = [r[0][:-len(".weight")] for r in rows if r[2] == "bn_w"]
        self.bn_off: Dict[str, int] = {}
        self.bn_c: Dict[str, int] = {}
        coff = 0
        for r in rows:
            if r[2] == "bn_w":
                name = r[0][:-len(".weight")]
                self.bn_off[name] = coff
                self.bn_c[name] = r[1][0]
                coff += _r4(r[1][0])
        self.bn_total = coff
        self.gamma_off = off
        self.beta_off = off + coff
        self.n_params = off + 2 * coff
        self.flat = torch.zeros(self.n_params, dtype=torch.float32, device=device)
        self.rmean = torch.zeros(coff, dtype=torch.float32, device=device)
        self.rvar = torch.ones(coff, dtype=torch.float32, device=device)
        self.nbt = torch.zeros(len(self.bn_names), dtype=torch.int64, device=device)
        self.fold_scale = torch.zeros(coff, dtype=torch.float32, device=device)
        self.fold_shift = torch.zeros(coff, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.n_params, dtype=torch.float32, device=device) if with_grads else None
        # transposed 1x1 weights for the input-gradient GEMM (student only)
        self.wt_off: Dict[str, int] = {}
        self.wt = None
        if with_grads:
            t = 0
            for key in self.order:
                e = self.entries[key]
                if e.kind == "pw":
                    self.wt_off[key] = t
                    t += _r4(e.n)
            self.wt = torch.zeros(t, dtype=torch.float32, device=device)

    # ---- views -------------------------------------------------------------------------
    def w(self, key: str) -> torch.Tensor:
        e = self.entries[key]
        return self.flat[e.off:e.off + e.n].view(e.native)

    def g(self, key: str) -> torch.Tensor:
        e = self.entries[key]
        return self.grad[e.off:e.off + e.n].view(e.native)

    def w_t(self, key: str) -> torch.Tensor:
        e = self.entries[key]
        o = self.wt_off[key]
        return self.wt[o:o + e.n].view(e.native[1], e.native[0])

    def bn(self, name: str):
        o, c = self.bn_off[name], self.bn_c[name]
        return {"C": c, "off": o,
                "gamma": self.flat[self.gamma_off + o:self.gamma_off + o + c],
                "beta": self.flat[self.beta_off + o:self.beta_off + o + c],
                "dgamma": None if self.grad is None else self.grad[self.gamma_off + o:self.gamma_off + o + c],
                "dbeta": None if self.grad is None else self.grad[self.beta_off + o:self.beta_off + o + c],
                "rmean": self.rmean[o:o + c], "rvar": self.rvar[o:o + c],
                "fscale": self.fold_scale[o:o + c], "fshift": self.fold_shift[o:o + c]}

    # ---- state dict ---------------------------------------------------------------------
    def load_state(self, state: Dict[str, torch.Tensor]) -> None:
        """Import a reference-layout state dict (missing keys raise, like load_state_dict(strict=True))."""
        host = torch.zeros(self.n_params, dtype=torch.float32)
        rm = torch.zeros(self.bn_total, dtype=torch.float32)
        rv = torch.ones(self.bn_total, dtype=torch.float32)
        nbt = torch.zeros(len(self.bn_names), dtype=torch.int64)
        for key, e in self.entries.items():
            t = state[key].detach().to("cpu", torch.float32)
            if tuple(t.shape) != e.shape:
                raise ValueError(f"shape mismatch for {key}: {tuple(t.shape)} vs {e.shape}")
            if e.kind == "dw" or (e.kind == "se_w" and e.native[0] != e.shape[0]):
                nat = t.reshape(e.shape[0], -1).t().contiguous()
            elif e.kind == "stem":
                nat = torch.zeros(e.native, dtype=torch.float32)
                nat[:, :e.shape[1] * 9] = t.reshape(e.shape[0], -1)
            else:
                nat = t.reshape(e.native)
            host[e.off:e.off + e.n] = nat.reshape(-1)
        for i, name in enumerate(self.bn_names):
            o, c = self.bn_off[name], self.bn_c[name]
            host[self.gamma_off + o:self.gamma_off + o + c] = state[name + ".weight"].detach().float().cpu()
            host[self.beta_off + o:self.beta_off + o + c] = state[name + ".bias"].detach().float().cpu()
            rm[o:o + c] = state[name + ".running_mean"].detach().float().cpu()
            rv[o:o + c] = state[name + ".running_var"].detach().float().cpu()
            k = name + ".num_batches_tracked"
            if k in state:
                nbt[i] = int(state[k])
        self.flat.copy_(host)
        self.rmean.copy_(rm)
        self.rvar.copy_(rv)
        self.nbt.copy_(nbt)

    def export_state(self) -> Dict[str, torch.Tensor]:
        host = self.flat.detach().cpu()
        rm, rv, nbt = self.rmean.cpu(), self.rvar.cpu(), self.nbt.cpu()
        out: Dict[str, torch.Tensor] = {}
        bn_index = {n: i for i, n in enumerate(self.bn_names)}
        for key, shape, kind in state_layout(self.spec):
            if key in self.entries:
                e = self.entries[key]
                nat = host[e.off:e.off + e.n].view(e.native)
                if e.kind == "dw" or (e.kind == "se_w" and e.native[0] != e.shape[0]):
                    t = nat.t().contiguous().view(e.shape)
                elif e.kind == "stem":
                    t = nat[:, :e.shape[1] * 9].contiguous().view(e.shape)
                else:
                    t = nat.reshape(e.shape).clone()
                out[key] = t
                continue
            name, leaf = key.rsplit(".", 1)
            o, c = self.bn_off[name], self.bn_c[name]
            if leaf == "weight":
                out[key] = host[self.gamma_off + o:self.gamma_off + o + c].clone()
            elif leaf == "bias":
                out[key] = host[self.beta_off + o:self.beta_off + o + c].clone()
            elif leaf == "running_mean":
                out[key] = rm[o:o + c].clone()
            elif leaf == "running_var":
                out[key] = rv[o:o + c].clone()
            else:
                out[key] = nbt[bn_index[name]].clone()
        return out

    def export_flat(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Any parameter-shaped flat buffer (Adam moments ...) as reference-layout tensors keyed by parameter name."""
        host = flat.detach().cpu()
        out: Dict[str, torch.Tensor] = {}
        for key, e in self.entries.items():
            nat = host[e.off:e.off + e.n].view(e.native)
            if e.kind == "dw" or (e.kind == "se_w" and e.native[0] != e.shape[0]):
                out[key] = nat.t().contiguous().view(e.shape)
            elif e.kind == "stem":
                out[key] = nat[:, :e.shape[1] * 9].contiguous().view(e.shape)
            else:
                out[key] = nat.reshape(e.shape).clone()
        for name in self.bn_names:
            o, c = self.bn_off[name], self.bn_c[name]
            out[name + ".weight"] = host[self.gamma_off + o:self.gamma_off + o + c].clone()
            out[name + ".bias"] = host[self.beta_off + o:self.beta_off + o + c].clone()
        return out

    def import_flat(self, flat: torch.Tensor, tensors: Dict[str, torch.Tensor]) -> None:
        host = torch.zeros(self.n_params, dtype=torch.float32)
        for key, e in self.entries.items():
            t = tensors[key].detach().float().cpu()
            if e.kind == "dw" or (e.kind == "se_w" and e.native[0] != e.shape[0]):
                nat = t.reshape(e.shape[0], -1).t().contiguous()
            elif e.kind == "stem":
                nat = torch.zeros(e.native, dtype=torch.float32)
                nat[:, :e.shape[1] * 9] = t.reshape(e.shape[0], -1)
            else:
                nat = t.reshape(e.native)
            host[e.off:e.off + e.n] = nat.reshape(-1)
        for name in self.bn_names:
            o, c = self.bn_off[name], self.bn_c[name]
            host[self.gamma_off + o:self.gamma_off + o + c] = tensors[name + ".weight"].float().cpu()
            host[self.beta_off + o:self.beta_off + o + c] = tensors[name + ".bias"].float().cpu()
        flat.copy_(host)

    def export_grads(self) -> Dict[str, torch.Tensor]:
        """Gradients in reference layout keyed like named_parameters() (tests / autograd facade)."""
        host = self.grad.detach().cpu()
        out: Dict[str, torch.Tensor] = {}
        for key, e in self.entries.items():
            nat = host[e.off:e.off + e.n].view(e.native)
            if e.kind == "dw" or (e.kind == "se_w" and e.native[0] != e.shape[0]):
                out[key] = nat.t().contiguous().view(e.shape)
            elif e.kind == "stem":
                out[key] = nat[:, :e.shape[1] * 9].contiguous().view(e.shape)
            else:
                out[key] = nat.reshape(e.shape).clone()
        for name in self.bn_names:
            o, c = self.bn_off[name], self.bn_c[name]
            out[name + ".weight"] = host[self.gamma_off + o:self.gamma_off + o + c].clone()
            out[name + ".bias"] = host[self.beta_off + o:self.beta_off + o + c].clone()
        return out


class Arena:
    """Bump allocator over a few big device chunks; reset() every step gives identical addresses each
    step (what a captured hipGraph needs) and avoids allocator traffic for ~10^3 activations."""

    def __init__(self, device, chunk_bytes: int = 1 << 30, zero_new: bool = False):
        self.device = device
        self.zero_new = zero_new        # accumulator arenas must start from zeros (atomics add into them)
        self.chunk_bytes = chunk_bytes
        self.chunks: List[torch.Tensor] = []
        self.ci = 0
        self.off = 0
        self.frozen = False

    def reset(self):
        self.ci = 0
        self.off = 0

    def used_bytes(self) -> int:
        return sum(c.numel() for c in self.chunks[:self.ci]) + self.off

    def alloc(self, shape, dtype=torch.float32) -> torch.Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        isz = torch.empty((), dtype=dtype).element_size()
        nbytes = (n * isz + 255) // 256 * 256
        while True:
            if self.ci == len(self.chunks):
                if self.frozen:
                    raise RuntimeError("arena would grow while frozen (a graph was captured on it)")
                mk = torch.zeros if self.zero_new else torch.empty
                self.chunks.append(mk(max(self.chunk_bytes, nbytes), dtype=torch.uint8, device=self.device))
            c = self.chunks[self.ci]
            if self.off + nbytes <= c.numel():
                t = c[self.off:self.off + n * isz].view(dtype).view(shape)
                self.off += nbytes
                return t
            self.ci += 1
            self.off = 0
