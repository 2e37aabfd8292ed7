"""Explicit forward / backward schedule of one EfficientDet on the HIP kernels (no autograd, no tracing).

The network is static, so the host simply issues the kernel sequence; the whole step is captured in a
hipGraph by step.py.  Activations are NHWC fp32 "rows" [B*H*W, C].  A conv's BatchNorm+swish is not
materialised: the conv writes its raw output z plus per-channel sums, `bn_finalize` turns the sums
into (scale, shift), and every CONSUMER applies act(z*scale+shift) while staging its input
(kernel prologue).  Only narrow tensors (MBConv block outputs, BiFPN node outputs) are materialised.

Reference schedule being restated: YetAnotherEfficientDet.forward (src/YetAnotherEfficientDet.py:662-685),
EfficientNet tap wrapper (:550-572), MBConvBlock.forward (src/YetAnotherEfficientNet.py:450-485),
BiFPN._forward_fast_attention (:320-392), Regressor/Classifier.forward (:463-532); the backward is
what autograd derives for them.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from .arch import NetSpec, BN_EPS, BN_MOMENTUM, pyramid_sizes
from .store import ParamStore, Arena

NONE, SWISH, SIGMOID = 0, 1, 2
import threading

_PACK = threading.local()      # .on: this thread is inside a pack forward (Net.forward(pack=...))
# entry points a pack forward may issue WITHOUT the group descriptor: they read no per-net parameters (or, the stems, are issued once per net)
PACK_PLAIN_OK = frozenset({"mmd_stem_conv_fwd", "mmd_maxpool_same_fwd", "mmd_memset_async"})


def call(name: str, *args):
    """_lib.call, except inside a pack forward: there every launch that reads parameters must go through Net._c (group descriptor set,
    honoured by the library or refused) - a plain launch would evaluate every net's images with the first net's weights, silently."""
    if getattr(_PACK, "on", False) and name not in PACK_PLAIN_OK:
        raise RuntimeError(f"{name} issued inside a pack forward without the group descriptor (engine.Net._c): it would read net 0's parameters for every image")
    return _lib.call(name, *args)



class WgLayer(ctypes.Structure):
    """MmdWgradLayer of include/mmdistill.h: one 1x1-conv weight gradient of the grouped launch (csrc/pw_wgrad_grouped.hip)."""
    _fields_ = [("dy", ctypes.c_void_p), ("x", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("in_scale", ctypes.c_void_p),
                ("in_shift", ctypes.c_void_p), ("gate", ctypes.c_void_p), ("M", ctypes.c_int), ("K", ctypes.c_int), ("N", ctypes.c_int),
                ("in_act", ctypes.c_int), ("rows_per_image", ctypes.c_int), ("mchunk", ctypes.c_int), ("nsplit", ctypes.c_int),
                ("ntn", ctypes.c_int), ("ntk", ctypes.c_int), ("item0", ctypes.c_int), ("tile0", ctypes.c_int), ("pad_", ctypes.c_int),
                ("ws_off", ctypes.c_longlong)]


# 1x1-conv weight gradients: deferred to the end of their backward segment and run as ONE persistent launch + a deterministic fold
# (MMD_NO_WG_GROUP=1: one launch per layer on the weight-gradient stream, fp32 atomics - the round-1 schedule)
WG_GROUP = not os.environ.get("MMD_NO_WG_GROUP")
# the OTHER leaves (depthwise / squeeze-excite / fusion-weight / bias gradients) stay forked off the main chain where they arise: small and
# latency-bound, they fill the main chain's gaps - deferring them to the segment end as well measured 20.8-22.1 ms/step against 20.5
WG_DEFER = WG_GROUP and bool(os.environ.get("MMD_WG_DEFER"))
# flush the grouped launch every WG_CHUNK recorded layers.  D2 records 99 layers per backward: 60 (round 2) = two launches, the second - every backbone layer below block 19 - alone at
# the very end of the backward (0.93 ms exposed); 45 = three, the last one only blocks 4..0 + stem while the second overlaps the high-resolution blocks' main chain
# (round 3, alternating runs: 16.52 / 16.81 -> 16.55 / 16.66 ms/step, medians 16.65 -> 16.47; 50: 16.70 / 16.82, 80: 16.99 / 17.01; round 2: 30 / 15 layers 20.5 / 20.8 vs 19.7 for 60)
# round 4 (alternating runs, four rounds): 45: 15.02 - 15.08, 46: 15.17 - 15.25, 47: 14.87 - 14.94, 48 / 49: 14.97 - 15.01 ms/step - the last flush is the step's exposed
# tail (rocprofv3: ~0.45 ms behind the main chain's last kernel at 45), 47 leaves it the two 256^2 project convs only; D4 and the bf16 modes do not care
WG_CHUNK = int(os.environ.get("MMD_WG_CHUNK", "47"))
# (dev) explicit flush points instead of the fixed stride: "47,80,94" = flush when the 47th, 80th, 94th layer of the backward has been recorded
WG_POINTS = frozenset(int(v) for v in os.environ.get("MMD_WG_POINTS", "").split(",") if v.strip())
WG_ROWS = int(os.environ.get("MMD_WG_ROWS", "4096"))      # rows per item: 4096 x (64x64 tile) measured best (1.33 ms per step vs 2.58 at 256)
WG_BLOCKS = int(os.environ.get("MMD_WG_GRID", "4096"))
# rows per item of a segment's LAST flush: it runs alone at the end of the backward (the thin 256^2 / 128^2 layers: 416 items of 4096 rows =
# 1.6 rounds of one block per CU, 278 us exposed); shorter items balance and keep more loads in flight
WG_ROWS_FINAL = int(os.environ.get("MMD_WG_ROWS_FINAL", "4096"))
# grid of a flush that runs BESIDE the backward's main chain (every flush but the last of a segment).  A chip-filling grid of these long-running
# blocks starves the chain's kernels of CU slots (round 4 trace: the two M131072 K24 N144 input-gradient GEMMs next to the second flush took
# 380 us apiece against 77 us alone); a thin persistent grid leaves the chain its slots and has ~2 ms of backward left to finish in
WG_BLOCKS_MID = int(os.environ.get("MMD_WG_GRID_MID", "4096"))


@dataclass
class Feat:
    """rows [B*H*W, C] plus the pending per-channel transform consumers must apply."""
    z: torch.Tensor
    B: int
    H: int
    W: int
    C: int
    scale: Optional[torch.Tensor] = None
    shift: Optional[torch.Tensor] = None
    act: int = NONE
    bn: Optional[tuple] = None     # train-mode forward: (stats[2C] double, gamma, beta, count) -> coefficients derived on the fly

    @property
    def M(self) -> int:
        return self.B * self.H * self.W


class GradSlot:
    """Gradient w.r.t. one activation tensor, accumulated over its consumers.  `remaining` counts the contributions still to come (the
    train forward counted the consumers); the kernel that writes the LAST one also produces the backward sums of the BatchNorm whose
    output the tensor is (`sums`, valid when `have_sums`), so that BatchNorm needs no reduce launch of its own."""
    def __init__(self, remaining: int = 0):
        self.t: Optional[torch.Tensor] = None
        self.remaining = remaining
        self.sums: Optional[torch.Tensor] = None
        self.have_sums = False
        self.linear_ok = True          # linear-sum mode (a tensor that is max-pooled somewhere): every contribution so far added its share
        self.stream: Optional[int] = None   # linear mode: the stream every contribution must be issued on (see Net._contrib)


@dataclass
class LazyDz:
    """A BatchNorm backward that is not materialised: the input- and weight-gradient GEMMs of the conv in front of the
    BatchNorm evaluate dz = BnBwd(g, z) in their operand prologues (mmd_pwconv_bwd_data_bn / _weight_bn)."""
    g: torch.Tensor
    z: torch.Tensor
    aff: tuple                  # (scale, shift, mean, invstd, ...)
    sums: torch.Tensor
    act: int
    mul_b: Optional[torch.Tensor]
    rpi: int
    bn_name: str
    count: int


LAZY_BN = not os.environ.get("MMD_NO_LAZY_BN")
# BatchNorm-backward reduce passes folded into the launches that complete their upstream gradient (MMD_NO_BNSUM_FOLD=1: the round-2
# schedule - a bn_bwd_reduce launch per BatchNorm and scale_acc launches for the skip / multi-consumer accumulations - for A/B timing)
FOLD_SUMS = not os.environ.get("MMD_NO_BNSUM_FOLD")
BN1_IN_DW = not os.environ.get("MMD_NO_BN1_IN_DW")    # MBConv: BatchNorm-1 backward in the depthwise input-gradient launch's prologue (no bn_bwd_apply, no dz1 tensor)
# BiFPN: the pooled operand's gradient scattered (fp32 atomics at the window arg-max) by the node backward launch itself, no dx tensor and no
# max-pool gather launch over the 4x larger source map; the BatchNorm sums of pooled tensors are then kept linearly (every contribution adds its share)
POOL_SCATTER = not os.environ.get("MMD_NO_POOL_SCATTER")
P5_IN_GEMM = not os.environ.get("MMD_NO_P5_IN_GEMM")  # MBConv: the pooled squeeze-excite / BN-1 backward pass in the project GEMM's epilogue (no chan_pool_bwd launch)
SE_WG_BATCH = not os.environ.get("MMD_NO_SE_WG_BATCH")
DW_S2_WG = not os.environ.get("MMD_NO_DW_S2_WG")         # ... and the conv's weight gradient (no dw_wgrad_kernel<k, 2> leaf)
DW_S2_SUMS = not os.environ.get("MMD_NO_DW_S2_SUMS")     # stride-2 depthwise input gradient takes the BatchNorm-0 backward sums (csrc/dwconv.hip dw_bwd_data_s2_sums_kernel)
# both squeeze-excite FC data gradients in one launch (mmd_se_fc_bwd_fused): correct, but every block recomputes its image's hidden gradient
# (S*C MACs, serial per wave) - measured 17.0 -> 17.7 ms/step against the two wide launches; off unless MMD_SE_FUSED=1
SE_FUSED = bool(os.environ.get("MMD_SE_FUSED"))
# stem weight gradient straight from the NCHW image (mmd_stem_conv_bwd_weight) instead of im2col + a GEMM item of the last grouped flush
STEM_WG_DIRECT = not os.environ.get("MMD_NO_STEM_WG_DIRECT")
# round 4: backward of the thin-input high-resolution expand convs (BatchNorm-0 backward, input gradient, weight gradient) in one pass over
# the 6x expanded gradient (csrc/mbconv_bwd_fused.hip): dz0 is neither stored nor read back, and the layer leaves the grouped weight-gradient
# launch at the exposed end of the backward.  Taken for layers with at least this many rows (MMD_NO_MBW=1: the two-GEMM form, for A/B timing)
# round 4: "lazy" BiFPN operands of the trainable net - a node's / down-channel conv's train-mode BatchNorm is applied by the CONSUMING node
# kernels while they load the operand (forward: from the live batch sums; backward: finalized coefficients) instead of by an mmd_affine_act
# launch behind every producer: 40 launches less on the student's forward chain (MMD_NO_LAZY_NODE=1: the materialising form, for A/B timing)
LAZY_NODE = not os.environ.get("MMD_NO_LAZY_NODE")
# round 4: whole-node BiFPN backward - the node's 1x1 conv's input gradient (BatchNorm backward in its operand prologue) runs inside the node
# backward launch (mmd_bifpn_node_bwd_full) instead of as a GEMM launch in front of it: 40 launches off the backward's serial chain
NODE_BWD_FULL = not os.environ.get("MMD_NO_NODE_BWD_FULL")
STEM_WG_SIDE = bool(os.environ.get("MMD_STEM_WG_SIDE"))      # (dev: the stem's weight gradient on the second side stream, the earlier placement)
# bf16 modes: the whole-node backward (whose 1x1 input-gradient product runs exact fp32 MFMA - closer to the fp32 reference than the mode's
# operand-rounding rule asks for, like the whole-node forward) up to width 160; D2 bf16 13.48 -> 13.25 ms/step, D4 (224) 49.9 -> 52.2: there
# the bf16 GEMM + two-launch form stays
NODE_FULL_BF16 = not os.environ.get("MMD_NO_NODE_FULL_BF16")
MBW_FUSED = not os.environ.get("MMD_NO_MBW")
MBW_MIN_ROWS = int(os.environ.get("MMD_MBW_MIN_ROWS", "32768"))      # squeeze-excite backward: one data-gradient launch per block, one weight-gradient launch per segment


def pack_nets(nets: List["Net"]) -> bool:
    """Lay the parameter stores of several frozen nets of one body architecture out at a constant stride inside shared buffers, so that
    they can be evaluated as ONE batch with per-group parameters (Net.forward(pack=...), csrc/common.h MmdGroup).  The nets must have been
    built with a common `stem_slot` (their stems may differ in input channels).  -> False when the stores do not line up."""
    p0 = nets[0].ps
    for n in nets[1:]:
        q = n.ps
        if n.trainable or q.n_params != p0.n_params or q.bn_total != p0.bn_total or q.order != p0.order or q.bn_off != p0.bn_off:
            return False
        if any(q.entries[k].off != p0.entries[k].off or (q.entries[k].native != p0.entries[k].native and q.entries[k].kind != "stem") for k in q.order):
            return False
    dev = p0.flat.device
    G = len(nets)
    flat = torch.zeros(G, p0.n_params, device=dev)
    fsc, fsh = torch.zeros(G, p0.bn_total, device=dev), torch.zeros(G, p0.bn_total, device=dev)
    for gi, n in enumerate(nets):
        flat[gi].copy_(n.ps.flat); fsc[gi].copy_(n.ps.fold_scale); fsh[gi].copy_(n.ps.fold_shift)
        n.ps.flat, n.ps.fold_scale, n.ps.fold_shift = flat[gi], fsc[gi], fsh[gi]      # (ParamStore derives every view from these on the fly)
    return True


class Net:
    def __init__(self, spec: NetSpec, device, trainable: bool, arena: Optional[Arena] = None,
                 zarena: Optional[Arena] = None, precision: str = "fp32", stem_slot: int = 0):
        if precision not in ("fp32", "bf16"):
            # ("bf16_hbm", bf16 STORAGE of the wide MBConv tensors, was deleted in round 6 together with its second library build: measured no
            # faster than "bf16" for three rounds - D4 / 768^2: 53.7 vs 53.6, 50.5 vs 49.2, 48.8 vs 48.4 ms/step - DESIGN.md section 5)
            raise ValueError(f"Unsupported precision {precision}")
        # "bf16" = mixed precision: the 1x1-conv GEMMs (forward, input- and weight-gradient) feed the bf16 MFMA, fp32
        # accumulate; tensors in HBM, depthwise convs, BatchNorm statistics, losses and Adam stay fp32
        self.precision = precision
        self._sfx = "_bf16" if self.precision == "bf16" else ""
        self.spec = spec
        self.device = device
        self.trainable = trainable
        self.ps = ParamStore(spec, device, with_grads=trainable, stem_slot=stem_slot)
        # grouped frozen nets (round 4, csrc/common.h MmdGroup): during a pack forward (n_groups, images_per_group, w_stride, bn_stride) -
        # every supported launch then covers the same layer of all the pack's nets, each workgroup reading its own net's parameters
        self._grp: Optional[tuple] = None
        self.arena = arena or Arena(device)
        self.zarena = zarena or Arena(device, 64 << 20, zero_new=True)       # per-step accumulators, zeroed in one memset
        self.bn_momentum = BN_MOMENTUM
        n = self.ps.bn_total
        if trainable:
            self.t_scale = torch.zeros(n, device=device)
            self.t_shift = torch.zeros(n, device=device)
            self.t_mean = torch.zeros(n, device=device)
            self.t_invstd = torch.zeros(n, device=device)
            self.stats_flat = torch.zeros(2 * n, dtype=torch.float64, device=device)
            lo = torch.zeros(n, dtype=torch.int32)
            lc = torch.zeros(n, dtype=torch.int32)
            for name in self.ps.bn_names:
                o, c = self.ps.bn_off[name], self.ps.bn_c[name]
                lo[o:o + c] = o
                lc[o:o + c] = c
            self.bn_layer_off = lo.to(device)
            self.bn_layer_c = lc.to(device)
            self.bn_count_host = torch.zeros(n, dtype=torch.float32)
            self.bn_count = torch.zeros(n, dtype=torch.float32, device=device)
            self._bn_count_key = None
            descs, tiles = [], 0
            for key in self.ps.wt_off:
                e = self.ps.entries[key]
                R, C = e.native
                descs.append([e.off, self.ps.wt_off[key], R, C, tiles])
                tiles += ((R + 31) // 32) * ((C + 31) // 32)
            self.wt_desc = torch.tensor(descs, dtype=torch.int64, device=device)
            self.wt_tiles = tiles
        self.tape: Dict[str, object] = {}
        self._slab_floats: Dict[tuple, int] = {}
        self._wg_read_done = None
        self._anchors: Dict[int, torch.Tensor] = {}
        self._side = None
        self._wg = None           # weight-gradient stream: nothing downstream of a wgrad until the optimizer
        self.mark_block, self.mark_event = -1, None
        self._wg_pending: list = []          # deferred 1x1-conv weight gradients of the current backward segment
        self._wg_count = 0                   # layers recorded so far in this backward
        self._leaf_pending: list = []        # other deferred leaves (closures), issued by _wg_flush
        self._wg_plans: Dict[tuple, dict] = {}  # (segment index, operand signature) -> planned table (built once: arena addresses repeat every step)
        self._wg_segment = 0
        self._linear: set = set()              # train forward: tensors that are a BiFPN node's POOLED operand (their gradient is scattered: linear sums)
        self._uses: Dict[int, int] = {}        # train forward: tensor -> number of gradient contributions its slot will receive
        self._bnout: Dict[int, tuple] = {}     # train forward: tensor y = BN(z) [* mul_b[image]] (+ skip) -> (z, mean, invstd, C, mul_b, rows_per_image)
        self._counting = False
        self._theta_desc: Dict[tuple, torch.Tensor] = {}
        self._se_wg: list = []                 # squeeze-excite FC weight gradients of the current backward segment (one batched launch)
        self._se_wg_tabs: Dict[tuple, torch.Tensor] = {}
        # test aid ("teacher forcing"): probe = dict that receives a copy of every materialised unit output of a forward (stem, MBConv
        # blocks, BiFPN down-channel convs and nodes); force_out = dict name -> rows [M, C] written over that output AFTER the probe, so the
        # next unit consumes the given tensor instead of this net's own (tests/test_gpu_net.py::test_teacher_forced_units_bf16)
        self.probe: Optional[Dict[str, torch.Tensor]] = None
        self.force_out: Optional[Dict[str, torch.Tensor]] = None
        self._lazy = False                     # this forward runs the BiFPN with lazy (consumer-applied) BatchNorms

    def _tf(self, name: str, t: torch.Tensor):
        if self.probe is not None:
            self.probe[name] = t.clone()
        if self.force_out is not None and name in self.force_out:
            t.copy_(self.force_out[name].view_as(t))

    def _use(self, f: "Feat"):
        if self._counting:
            k = f.z.data_ptr()
            self._uses[k] = self._uses.get(k, 0) + 1

    def _c(self, name: str, *args):
        """call() for the frozen-forward entry points that honour the group mode: inside a pack forward the group is set around the launch."""
        if self._grp is None:
            return call(name, *args)
        dll = _lib.LIB.load()
        if dll.mmd_set_group(*self._grp) != 0:
            raise RuntimeError("mmd_set_group refused %r" % (self._grp,))
        ok = False
        try:
            rc = _lib.call(name, *args)
            ok = True
        finally:
            # -22 from the clearing call: no launch read the descriptor, i.e. `name` has no group mode and ran with net 0's parameters
            if dll.mmd_set_group(1, 0, 0, 0) != 0 and ok:
                raise RuntimeError(f"{name} does not honour the group descriptor (csrc/common.h MmdGroup): refused inside a pack forward")
        return rc

    # ------------------------------------------------------------------ parameters
    def load_state(self, state):
        self.ps.load_state(state)
        self.refresh()

    def refresh(self):
        """Recompute derived buffers after parameters changed: eval-BN fold and transposed 1x1 weights."""
        ps = self.ps
        if not ps.flat.is_cuda:
            return            # host-side construction (state-dict plumbing, CPU tests): nothing to derive yet
        call("mmd_bn_fold", ps.flat[ps.gamma_off:ps.gamma_off + ps.bn_total], ps.flat[ps.beta_off:ps.beta_off + ps.bn_total],
             ps.rmean, ps.rvar, BN_EPS, ps.fold_scale, ps.fold_shift, ps.bn_total)
        if self.trainable:
            self.refresh_wt()

    def refresh_wt(self):
        ps = self.ps
        call("mmd_transpose_batched", ps.flat, ps.wt, self.wt_desc, self.wt_desc.shape[0], self.wt_tiles)

    # ------------------------------------------------------------------ small helpers
    def _alloc(self, *shape):
        return self.arena.alloc(shape)

    def _zalloc(self, shape, dtype=torch.float32):
        return self.zarena.alloc(shape, dtype)

    def begin_step(self):
        """Reset the bump arenas and zero the accumulator arena (one memset for all per-step sums)."""
        self.arena.reset()
        used = self.zarena.used_bytes()
        self.zarena.reset()
        for c in self.zarena.chunks:
            call("mmd_memset_async", c, 0, c.numel())
        if self.trainable:
            call("mmd_memset_async", self.stats_flat, 0, self.stats_flat.numel() * 8)
        self.tape = {}

    def _bn_stats(self, name: str, train: bool):
        """Raw-sum accumulator [2C] (double) of BN `name` for this step (train) or None (eval)."""
        if not train:
            return None
        b = self.ps.bn(name)
        o, c = b["off"], b["C"]
        return self.stats_flat[2 * o:2 * o + 2 * c]

    def _bn_aff(self, name: str, train: bool, stats, count: int):
        """-> (scale, shift, mean, invstd, live).  eval: folded running stats.  train: views that the batched finalize
        at the end of the forward fills (the backward reads them) + the `live` tuple forward consumers use."""
        b = self.ps.bn(name)
        if not train:
            return b["fscale"], b["fshift"], None, None, None
        o, c = b["off"], b["C"]
        self.bn_count_host[o:o + c] = float(count)
        return (self.t_scale[o:o + c], self.t_shift[o:o + c], self.t_mean[o:o + c], self.t_invstd[o:o + c],
                (stats, b["gamma"], b["beta"], int(count)))

    @staticmethod
    def _xf(x: Feat):
        """prologue arguments (scale, shift, act, stats, gamma, beta, count) of a forward consumer of x"""
        if x.bn is not None:
            return (None, None, x.act, x.bn[0], x.bn[1], x.bn[2], x.bn[3])
        return (x.scale, x.shift, x.act, None, None, None, 0)

    STATS_SLOTS = 0 if os.environ.get("MMD_NO_SLOTS") else 64
    DW_WG = not os.environ.get("MMD_NO_DW_WG")          # stride-1 backbone layers: depthwise weight gradient inside the input-gradient launch
    NODE_WG = not os.environ.get("MMD_NO_NODE_WG")      # BiFPN nodes: depthwise weight gradient inside the node's backward launch
    FUSE_NODE_TRAIN = not os.environ.get("MMD_NO_NODE_FUSE_TRAIN")   # trainable net: a BiFPN node's forward (fusion, depthwise, 1x1 conv, BN sums) in one kernel
    FUSE_NODE = not os.environ.get("MMD_NO_NODE_FUSE")  # frozen nets: a BiFPN node (fusion, depthwise, 1x1 conv, BN) in one kernel
    FUSE_FRONT = not os.environ.get("MMD_NO_MBX")       # frozen nets: expand + depthwise of the thin-input blocks in one kernel
    # widths above 160 (D4's 224): the whole-node kernel holds two 8x8-pixel tiles in 157 KB of LDS (one block per CU) and reads its 1x1
    # weights from L2 - a win on the latency-bound small maps, a loss on the large ones, where the two-launch path runs a chip-filling GEMM
    # (round 4, D4 / 768^2 at B = 8: every level fused 56.7 ms/step against 54.0 unfused)
    NODE_FUSE_WIDE_MAXROWS = int(os.environ.get("MMD_NODE_FUSE_WIDE_MAXROWS", "1152"))      # (neutral there: fewer launches, same time; 4608: +0.4 ms; every level: +2.7 ms)

    def _node_fusable(self, in0: "Feat") -> bool:
        if not self.ps.flat.is_cuda or _lib.LIB.load().mmd_bifpn_node_fused_supported(in0.C) != 1:
            return False
        return in0.C <= 160 or in0.M <= self.NODE_FUSE_WIDE_MAXROWS

    def _stats_ws(self, stats, M: int, C: int):
        """(workspace, slots) for a BatchNorm-sum producer over M rows: the thin full-resolution layers would send
        thousands of same-address f64 atomics (memory-side, ~17 ns each) to 2C addresses; the kernels spread them over
        `slots` zeroed copies and fold.  Only worth a buffer when M is large (the C side decides whether to use it)."""
        if stats is None or M < 16384 or not self.STATS_SLOTS:
            return None, 0
        return self._zalloc((self.STATS_SLOTS * 2 * C,), torch.float64), self.STATS_SLOTS

    def _pw(self, x: Feat, wkey: str, N: int, bias=None, stats=None, out_aff=None, out_act=NONE, residual=None,
            gate=None, y=None, ybs=0, yoff=0, plain_in=False):
        M, K = x.M, x.C
        if y is None:
            y = self._alloc(M, N)
        xf = (None, None, NONE, None, None, None, 0) if plain_in else self._xf(x)
        args = (x.z, self.ps.w(wkey), y, M, K, N, *xf, gate, x.H * x.W, bias, out_aff[0] if out_aff else None, out_aff[1] if out_aff else None, out_act,
                residual, stats, ybs, yoff, *self._stats_ws(stats, M, N))
        if self._grp is None and not self._sfx and not plain_in and ybs == 0:
            # (fp32, outside a pack forward, with a producer transform: the launch may take the all-N K-sliced slab kernel, csrc/pw_slab.hip,
            # which needs a workspace for its K slices' partial slabs when the launch has few row slabs)
            call("mmd_pwconv_fwd_form", *args, *self._slab_ws(M, K, N, 0), 0)
        else:
            self._c("mmd_pwconv_fwd" + self._sfx, *args)
        return y

    def _slab_ws(self, M: int, K: int, N: int, bn_operand: int):
        """(workspace, floats) for the K slices of a slab-kernel launch of this shape (mmd_pwconv_slab_ws_floats; (None, 0): the launch runs
        unsliced or on another kernel family).  Plain arena memory: the partial slabs need no zeroing."""
        if not self.ps.flat.is_cuda:
            return None, 0
        key = (M, K, N, bn_operand)
        n = self._slab_floats.get(key)
        if n is None:
            n = self._slab_floats[key] = int(_lib.LIB.load().mmd_pwconv_slab_ws_floats(M, K, N, bn_operand))
        return (self._alloc(n), n) if n > 0 else (None, 0)

    def _dw(self, x: Feat, wkey: str, k: int, s: int, stats=None, out_aff=None, out_act=NONE, pool=None):
        OH, OW = -(-x.H // s), -(-x.W // s)
        y = self._alloc(x.B * OH * OW, x.C)
        args = (x.z, self.ps.w(wkey), y, x.B, x.H, x.W, x.C, k, s, *self._xf(x),
                out_aff[0] if out_aff else None, out_aff[1] if out_aff else None, out_act, stats, pool,
                *self._stats_ws(stats, x.B * OH * OW, x.C))
        self._c("mmd_dwconv_fwd", *args)
        return y, OH, OW

    def anchors(self, image_size: int) -> torch.Tensor:
        """[A,4] (y1,x1,y2,x2) fp32, float64 math then cast (src/YetAnotherEfficientDet.py:116-147)."""
        if image_size not in self._anchors:
            import itertools
            import numpy as np
            scales = np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)])
            ratios = [(1.0, 1.0), (1.4, 0.7), (0.7, 1.4)]
            allb = []
            for stride in [8, 16, 32, 64, 128]:
                lv = []
                for scale, ratio in itertools.product(scales, ratios):
                    base = self.spec.anchor_scale * stride * scale
                    ax, ay = base * ratio[0] / 2.0, base * ratio[1] / 2.0
                    x = np.arange(stride / 2, image_size, stride)
                    xv, yv = np.meshgrid(x, x)
                    xv, yv = xv.reshape(-1), yv.reshape(-1)
                    b = np.vstack((yv - ay, xv - ax, yv + ay, xv + ax)).swapaxes(0, 1)
                    lv.append(np.expand_dims(b, 1))
                allb.append(np.concatenate(lv, axis=1).reshape(-1, 4))
            self._anchors[image_size] = torch.from_numpy(
                np.ascontiguousarray(np.vstack(allb).astype(np.float32))).contiguous().to(self.device)
        return self._anchors[image_size]

    # ------------------------------------------------------------------ forward
    def forward(self, x, train: bool = False, drop_scale: Optional[torch.Tensor] = None,
                raw_logits: bool = False, pack: Optional[list] = None):
        """x: [B,Cin,S,S] NCHW fp32 on device.  drop_scale: [n_skip_blocks, B] = mask/keep (train only).
        Returns cls [B,A,NC] (probabilities; pre-sigmoid logits with raw_logits=True, a calibration aid), reg [B,A,4],
        features: list of 5 Feat (NHWC rows)."""
        if train and not self.trainable:
            raise RuntimeError("train-mode forward on a frozen (teacher) net")
        spec, ps = self.spec, self.ps
        if pack is not None:
            # pack forward: `pack` = the frozen nets evaluated together (this one first), x = their inputs [Bg, Cin_g, S, S] in the same order.
            # Everything behind the stems runs ONCE over the G x Bg images with per-group parameters (pack_nets laid the stores out alike)
            assert not train and pack[0] is self and len(pack) == len(x) and len(pack) > 1
            Bg, S = x[0].shape[0], x[0].shape[2]
            assert all(xi.shape[0] == Bg and xi.shape[2] == S for xi in x)
            self._grp = (len(pack), Bg, ps.n_params, ps.bn_total)
            _PACK.on = True
            try:
                return self._forward(x, train, drop_scale, raw_logits, pack)
            finally:
                self._grp = None
                _PACK.on = False
        return self._forward(x, train, drop_scale, raw_logits, None)

    def _forward(self, x, train, drop_scale, raw_logits, pack):
        spec, ps = self.spec, self.ps
        if pack is not None:
            B, S = x[0].shape[0] * len(pack), x[0].shape[2]
        else:
            B, Cin, S, _ = x.shape
        tape = self.tape if train else {}
        self._counting = train
        if train:
            self._uses, self._bnout, self._linear = {}, {}, set()
        P = "backbone_net.model"
        # ---- stem: im2col + GEMM
        OH = (S + 1) // 2
        # direct 3x3/s2 stem conv (NCHW image -> NHWC rows); the im2col matrix is only built in the backward, for the
        # weight gradient, on the wgrad stream
        st = self._bn_stats(f"{P}._bn0", train)
        sc, sh, mu, istd, live = self._bn_aff(f"{P}._bn0", train, st, B * OH * OH)
        z = self._alloc(B * OH * OH, spec.stem_out)
        wstem = ps.w(f"{P}._conv_stem.conv.weight")
        if train:
            call("mmd_stem_conv_fwd", x, wstem, z, B, Cin, S, S, ps.stem_kp, spec.stem_out, None, None, NONE, st,
                 *self._stats_ws(st, B * OH * OH, spec.stem_out))
            cur = Feat(z, B, OH, OH, spec.stem_out, sc, sh, SWISH, live)
            tape["stem"] = (x, cur, mu, istd)
        elif pack is not None:
            # the stems differ (input channels): one launch per net, each writing its images' rows of the shared stem output
            Bg, rows = x[0].shape[0], x[0].shape[0] * OH * OH
            for gi, (net, xg) in enumerate(zip(pack, x)):
                b0 = net.ps.bn(f"{P}._bn0")
                call("mmd_stem_conv_fwd", xg, net.ps.w(f"{P}._conv_stem.conv.weight"), z[gi * rows:(gi + 1) * rows], Bg, xg.shape[1], S, S,
                     net.ps.stem_kp, spec.stem_out, b0["fscale"], b0["fshift"], SWISH, None, None, 0)
            cur = Feat(z, B, OH, OH, spec.stem_out)
        else:
            # frozen net: the folded BN + swish ride in the producer's epilogue (once per element) instead of the depthwise
            # prologue, which would redo them for every halo pixel (1.6x for 3x3, 2.25x for 5x5 tiles)
            call("mmd_stem_conv_fwd", x, wstem, z, B, Cin, S, S, ps.stem_kp, spec.stem_out, sc, sh, SWISH, None, None, 0)
            cur = Feat(z, B, OH, OH, spec.stem_out)
            self._tf("stem", z)
        taps: List[Feat] = []
        skip_i = 0
        for blk in spec.blocks:
            q = f"{P}._blocks.{blk.idx}"
            inp = cur
            rec = {"inp": inp}
            self._use(inp)                  # the block's own path ...
            if blk.skip:
                self._use(inp)              # ... and the identity skip
            fused_front = (not train and blk.expand != 1 and self.FUSE_FRONT and ps.flat.is_cuda
                           and _lib.LIB.load().mmd_mbconv_expand_dw_supported(inp.C, blk.cmid, blk.kernel, blk.stride) == 1)
            if fused_front:
                f0 = None           # frozen net, thin input: expand + depthwise in one kernel below, the expanded tensor stays in LDS
            elif blk.expand != 1:
                st0 = self._bn_stats(f"{q}._bn0", train)
                a0 = self._bn_aff(f"{q}._bn0", train, st0, inp.M)
                if train:
                    z0 = self._pw(inp, f"{q}._expand_conv.conv.weight", blk.cmid, stats=st0)
                    f0 = Feat(z0, B, inp.H, inp.W, blk.cmid, a0[0], a0[1], SWISH, a0[4])
                else:           # frozen net: activate in the producer's epilogue (see the stem)
                    z0 = self._pw(inp, f"{q}._expand_conv.conv.weight", blk.cmid, out_aff=(a0[0], a0[1]), out_act=SWISH)
                    f0 = Feat(z0, B, inp.H, inp.W, blk.cmid)
                rec["f0"], rec["bn0"] = f0, a0
            else:
                f0 = inp
            res = inp.z if blk.skip else None
            H1, W1 = -(-inp.H // blk.stride), -(-inp.W // blk.stride)
            M1 = B * H1 * W1
            # (frozen nets: Q36 fixed-point integer sums - bit-reproducible whatever order the blocks' atomics arrive in, csrc/common.h mmd_pool_add)
            pooled = self._zalloc((B, blk.cmid)) if train else self._zalloc((B, blk.cmid), torch.int64)
            hpre = self._alloc(B, blk.se)
            gate = self._alloc(B, blk.cmid)
            se_w = (ps.w(f"{q}._se_reduce.conv.weight"), ps.w(f"{q}._se_reduce.conv.bias"),
                    ps.w(f"{q}._se_expand.conv.weight"), ps.w(f"{q}._se_expand.conv.bias"))
            if train:
                st1 = self._bn_stats(f"{q}._bn1", True)
                z1, _, _ = self._dw(f0, f"{q}._depthwise_conv.conv.weight", blk.kernel, blk.stride, stats=st1)
                a1 = self._bn_aff(f"{q}._bn1", True, st1, M1)
                f1 = Feat(z1, B, H1, W1, blk.cmid, a1[0], a1[1], SWISH, a1[4])
                call("mmd_chan_pool", z1, *self._xf(f1)[:2], *self._xf(f1)[3:], SWISH, None, pooled, 1.0 / (H1 * W1), B,
                     H1 * W1, blk.cmid)
                call("mmd_se_fc_fwd", pooled, *se_w, hpre, gate, B, blk.cmid, blk.se)
                st2 = self._bn_stats(f"{q}._bn2", True)
                z2 = self._pw(f1, f"{q}._project_conv.conv.weight", blk.cout, stats=st2, gate=gate)
                a2 = self._bn_aff(f"{q}._bn2", True, st2, M1)
                y = self._alloc(M1, blk.cout)
                rs = None
                if blk.skip and blk.drop_rate and drop_scale is not None:
                    rs = drop_scale[skip_i]
                call("mmd_affine_act", z2, None, None, a2[4][0], a2[4][1], a2[4][2], a2[4][3], NONE, rs, H1 * W1, res, y, M1,
                     blk.cout)
                rec.update(f1=f1, bn1=a1, pooled=pooled, hpre=hpre, gate=gate, z2=z2, bn2=a2, rs=rs)
                self._bnout[y.data_ptr()] = (z2, a2[2], a2[3], blk.cout, rs, H1 * W1)
            else:
                # frozen net: BN1+swish and the SE average pool ride in the depthwise epilogue (no separate pool pass)
                b1 = ps.bn(f"{q}._bn1")
                if fused_front:
                    b0 = ps.bn(f"{q}._bn0")
                    a1v = self._alloc(M1, blk.cmid)
                    self._c("mmd_mbconv_expand_dw_fwd", inp.z, ps.w(f"{q}._expand_conv.conv.weight"), b0["fscale"], b0["fshift"],
                         ps.w(f"{q}._depthwise_conv.conv.weight"), b1["fscale"], b1["fshift"], a1v, pooled, B, inp.H, inp.W, inp.C,
                         blk.cmid, blk.kernel, blk.stride)
                else:
                    a1v, _, _ = self._dw(f0, f"{q}._depthwise_conv.conv.weight", blk.kernel, blk.stride,
                                         out_aff=(b1["fscale"], b1["fshift"]), out_act=SWISH, pool=pooled)
                f1 = Feat(a1v, B, H1, W1, blk.cmid)
                self._c("mmd_se_fc_fwd_q", pooled, *se_w, hpre, gate, B, blk.cmid, blk.se)
                b2 = ps.bn(f"{q}._bn2")
                y = self._pw(f1, f"{q}._project_conv.conv.weight", blk.cout, gate=gate,
                             out_aff=(b2["fscale"], b2["fshift"]), residual=res)
            if blk.skip:
                skip_i += 1
            if blk.idx == self.mark_block and self.ps.flat.is_cuda:
                self.mark_event = torch.cuda.current_stream().record_event()      # "this net is past block k" (step.py staggers the teachers on it)
            cur = Feat(y, B, H1, W1, blk.cout)
            self._tf(f"blk{blk.idx}", y)
            rec["out"] = cur
            if train:
                tape[f"blk{blk.idx}"] = rec
            if blk.idx in spec.taps:
                taps.append(cur)
        # ---- BiFPN
        feats = self._bifpn(taps, train, tape)
        # ---- heads
        for f in feats:
            self._use(f)                    # one contribution: heads' input gradients (+ the MTA loss' feature gradient) in one launch
        A = sum(f.H * f.W for f in feats) * spec.num_anchors
        reg = self._alloc(B, A, 4)
        cls = self._alloc(B, A, spec.num_classes)
        self._head("regressor", feats, 4, reg, A, NONE, train, tape)
        self._head("classifier", feats, spec.num_classes, cls, A, NONE if raw_logits else SIGMOID, train, tape)
        if train:
            tape["feats"] = feats
            tape["A"] = A
            tape["cls"] = cls
            self._counting = False
            # every BN layer finalized in one launch: running stats + (scale, shift, mean, invstd) for the backward
            key = (B, S)
            if self._bn_count_key != key:
                self.bn_count.copy_(self.bn_count_host)
                self._bn_count_key = key
            call("mmd_bn_finalize_all", self.stats_flat, self.bn_count, self.bn_layer_off, self.bn_layer_c,
                 ps.flat[ps.gamma_off:ps.gamma_off + ps.bn_total], ps.flat[ps.beta_off:ps.beta_off + ps.bn_total],
                 ps.rmean, ps.rvar, float(self.bn_momentum), BN_EPS, self.t_scale, self.t_shift, self.t_mean,
                 self.t_invstd, ps.bn_total, ps.nbt, ps.nbt.numel())      # (+ num_batches_tracked += 1 for every BN, in the same launch)
        return cls, reg, feats

    def _sep_bn(self, name: str, x: Feat, train: bool, rec: dict, bn_name: Optional[str] = None, y=None, zd=None) -> Feat:
        """SeparableConvBlock(norm=True, activation=False) on a materialised input -> materialised output.
        zd: depthwise output already produced by the fused fusion+depthwise kernel."""
        ps = self.ps
        W = x.C
        if zd is None:
            zd, _, _ = self._dw(x, f"{name}.depthwise_conv.conv.weight", 3, 1)
        zdf = Feat(zd, x.B, x.H, x.W, W)
        bn_name = bn_name or f"{name}.bn"
        bias = ps.w(f"{name}.pointwise_conv.conv.bias")
        if train:
            st = self._bn_stats(bn_name, True)
            z = self._pw(zdf, f"{name}.pointwise_conv.conv.weight", W, bias=bias, stats=st)
            a = self._bn_aff(bn_name, True, st, x.M)
            if y is None:
                y = self._alloc(x.M, W)
            call("mmd_affine_act", z, None, None, a[4][0], a[4][1], a[4][2], a[4][3], NONE, None, 0, None, y, x.M, W)
            rec.update(zd=zdf, z=z, bn=a)
            self._bnout[y.data_ptr()] = (z, a[2], a[3], W, None, 0)
        else:
            b = ps.bn(bn_name)
            y = self._pw(zdf, f"{name}.pointwise_conv.conv.weight", W, bias=bias, out_aff=(b["fscale"], b["fshift"]), y=y)
        return Feat(y, x.B, x.H, x.W, W)

    def _down_channel(self, name: str, x: Feat, train: bool, tape: dict, lazy: bool = False) -> Feat:
        ps = self.ps
        W = self.spec.fpn_w
        bias = ps.w(f"{name}.0.conv.bias")
        rec = {"x": x}
        if train and lazy:
            # lazy output: the raw conv output + its BatchNorm as a pending transform the consuming node kernels apply (no affine launch)
            st = self._bn_stats(f"{name}.1", True)
            z = self._pw(x, f"{name}.0.conv.weight", W, bias=bias, stats=st)
            a = self._bn_aff(f"{name}.1", True, st, x.M)
            rec.update(z=z, bn=a)
            tape[name] = rec
            self._use(x)
            self._bnout[z.data_ptr()] = (z, a[2], a[3], W, None, 0)
            out = Feat(z, x.B, x.H, x.W, W, a[0], a[1], NONE, a[4])
            rec["out"] = out
            return out
        if train:
            st = self._bn_stats(f"{name}.1", True)
            z = self._pw(x, f"{name}.0.conv.weight", W, bias=bias, stats=st)
            a = self._bn_aff(f"{name}.1", True, st, x.M)
            y = self._alloc(x.M, W)
            call("mmd_affine_act", z, None, None, a[4][0], a[4][1], a[4][2], a[4][3], NONE, None, 0, None, y, x.M, W)
            rec.update(z=z, bn=a)
            tape[name] = rec
            self._use(x)
            self._bnout[y.data_ptr()] = (z, a[2], a[3], W, None, 0)
        else:
            b = ps.bn(f"{name}.1")
            y = self._pw(x, f"{name}.0.conv.weight", W, bias=bias, out_aff=(b["fscale"], b["fshift"]))
        out = Feat(y, x.B, x.H, x.W, W)
        self._tf(name, y)
        rec["out"] = out
        return out

    def _pool(self, x: Feat) -> Feat:
        OH, OW = (x.H + 1) // 2, (x.W + 1) // 2
        y = self._alloc(x.B * OH * OW, x.C)
        call("mmd_maxpool_same_fwd", x.z, y, x.B, x.H, x.W, x.C)
        self._use(x)
        return Feat(y, x.B, OH, OW, x.C)

    def _node(self, cell: str, conv: str, theta: str, in0: Feat, in1: Optional[Feat], up: Optional[Feat],
              pl: Optional[Feat], train: bool, tape: dict, y=None) -> Feat:
        out = self._node_impl(cell, conv, theta, in0, in1, up, pl, train, tape, y)
        self._tf(f"{cell}.{conv}", out.z)
        return out

    def _node_impl(self, cell: str, conv: str, theta: str, in0: Feat, in1: Optional[Feat], up: Optional[Feat],
                   pl: Optional[Feat], train: bool, tape: dict, y=None) -> Feat:
        th = self.ps.w(f"{cell}.{theta}")
        if not train and self.FUSE_NODE and self._node_fusable(in0):
            # frozen net: fusion + depthwise + 1x1 conv + folded BN in one launch, the depthwise output never leaves the CU
            name = f"{cell}.{conv}"
            b = self.ps.bn(f"{name}.bn")
            if y is None:
                y = self._alloc(in0.M, in0.C)
            self._c("mmd_bifpn_node_fwd_fused", in0.z, in1.z if in1 else None, up.z if up else None, pl.z if pl else None, th,
                 self.ps.w(f"{name}.depthwise_conv.conv.weight"), self.ps.w(f"{name}.pointwise_conv.conv.weight"),
                 self.ps.w(f"{name}.pointwise_conv.conv.bias"), b["fscale"], b["fshift"], y, in0.B, in0.H, in0.W, in0.C)
            return Feat(y, in0.B, in0.H, in0.W, in0.C)
        # (also in the bf16 modes: the node's 1x1 conv then runs exact fp32 products in the forward - closer to the fp32 reference than the
        # mode's operand-rounding rule asks for; its two gradient GEMMs follow the mode)
        if train and self.FUSE_NODE_TRAIN and self.NODE_WG and self._node_fusable(in0):
            # trainable net: fusion + depthwise + 1x1 conv + BatchNorm sums in one launch (raw z out, depthwise output kept for the backward)
            name, W = f"{cell}.{conv}", in0.C
            bn_name = f"{name}.bn"
            st = self._bn_stats(bn_name, True)
            z, zd = self._alloc(in0.M, W), self._alloc(in0.M, W)
            ops = (in0, in1, up, pl)
            lz = [o is not None and o.bn is not None for o in ops]          # operands whose BatchNorm is still pending (lazy producers)
            args = (in0.z, in1.z if in1 else None, up.z if up else None, pl.z if pl else None, th,
                    self.ps.w(f"{name}.depthwise_conv.conv.weight"), self.ps.w(f"{name}.pointwise_conv.conv.weight"),
                    self.ps.w(f"{name}.pointwise_conv.conv.bias"), z, zd, st, in0.B, in0.H, in0.W, W)
            if any(lz):
                assert self._lazy and all(o.act == NONE for o, l in zip(ops, lz) if l)
                vp = ctypes.c_void_p
                pick = lambda k: (vp * 4)(*[(o.bn[k].data_ptr() if l else None) for o, l in zip(ops, lz)])
                cnt = (ctypes.c_longlong * 4)(*[(int(o.bn[3]) if l else 0) for o, l in zip(ops, lz)])
                call("mmd_bifpn_node_fwd_fused_train_lz", *args, pick(0), pick(1), pick(2), cnt)
            else:
                call("mmd_bifpn_node_fwd_fused_train", *args)
            a = self._bn_aff(bn_name, True, st, in0.M)
            for operand in ops:
                if operand is not None:
                    self._use(operand)
            if pl is not None:
                self._linear.add(pl.z.data_ptr())
            zdf = Feat(zd, in0.B, in0.H, in0.W, W)
            if self._lazy and y is None:
                # lazy output (every cell but the last, whose outputs the heads and the MTA loss read): no affine launch
                out = Feat(z, in0.B, in0.H, in0.W, W, a[0], a[1], NONE, a[4])
                self._bnout[z.data_ptr()] = (z, a[2], a[3], W, None, 0)
            else:
                if y is None:
                    y = self._alloc(in0.M, W)
                call("mmd_affine_act", z, None, None, a[4][0], a[4][1], a[4][2], a[4][3], NONE, None, 0, None, y, in0.M, W)
                out = Feat(y, in0.B, in0.H, in0.W, W)
                self._bnout[y.data_ptr()] = (z, a[2], a[3], W, None, 0)
            tape.setdefault(cell + ".nodes", []).append({"in0": in0, "in1": in1, "up": up, "pl": pl, "theta": theta, "f": zdf, "conv": conv,
                                                         "zd": zdf, "z": z, "bn": a, "out": out})
            return out
        # the fused activation is not materialised: the node's backward launch recomputes it, also for the depthwise weight gradient
        # (MMD_NO_NODE_WG=1: the earlier form - f written here, weight gradient by its own launch on the side stream - for A/B timing)
        f = self._alloc(in0.M, in0.C) if (train and not self.NODE_WG) else None
        zd = self._alloc(in0.M, in0.C)
        self._c("mmd_bifpn_node_dw_fwd", in0.z, in1.z if in1 else None, up.z if up else None, pl.z if pl else None, th,
                self.ps.w(f"{cell}.{conv}.depthwise_conv.conv.weight"), f, zd, in0.B, in0.H, in0.W, in0.C)
        ff = Feat(f if f is not None else zd, in0.B, in0.H, in0.W, in0.C)
        for operand in (in0, in1, up, pl):
            if operand is not None:
                self._use(operand)
        if pl is not None and train:
            self._linear.add(pl.z.data_ptr())
        rec = {"in0": in0, "in1": in1, "up": up, "pl": pl, "theta": theta, "f": ff, "conv": conv}
        out = self._sep_bn(f"{cell}.{conv}", ff, train, rec, y=y, zd=zd)
        rec["out"] = out
        if train:
            tape.setdefault(cell + ".nodes", []).append(rec)
        return out

    def _bifpn(self, taps: List[Feat], train: bool, tape: dict) -> List[Feat]:
        feats: List[Feat] = list(taps)
        # lazy operands need every node on the whole-node train kernel (widths <= 160), the scatter form of the pooled operand's gradient
        # (the gather launch reads operand values) and no test hooks that read / overwrite materialised outputs
        self._lazy = (train and LAZY_NODE and self.FUSE_NODE_TRAIN and self.NODE_WG and FOLD_SUMS and POOL_SCATTER and self.ps.flat.is_cuda
                      and self.probe is None and self.force_out is None and self.spec.fpn_w <= 160
                      and _lib.LIB.load().mmd_bifpn_node_fused_supported(self.spec.fpn_w) == 1)
        lz = self._lazy
        for c in range(self.spec.fpn_cells):
            cell = f"bifpn.{c}"
            if c == 0:
                p3, p4, p5 = feats
                c6 = self._down_channel(f"{cell}.p5_to_p6", p5, train, tape)         # (max-pooled by its own launch: materialised)
                p6_in = self._pool(c6)
                p7_in = self._pool(p6_in)
                p3_in = self._down_channel(f"{cell}.p3_down_channel", p3, train, tape, lazy=lz)
                p4_in = self._down_channel(f"{cell}.p4_down_channel", p4, train, tape, lazy=lz)
                p5_in = self._down_channel(f"{cell}.p5_down_channel", p5, train, tape, lazy=lz)
                if train:
                    tape[cell + ".first"] = {"c6": c6, "p6_in": p6_in, "p7_in": p7_in}
            else:
                p3_in, p4_in, p5_in, p6_in, p7_in = feats
            last = c == self.spec.fpn_cells - 1
            if last:    # final outputs land in ONE pyramid row buffer so the shared-weight heads run all levels per launch
                sizes = [(p3_in.H, p3_in.W), (p4_in.H, p4_in.W), (p5_in.H, p5_in.W), (p6_in.H, p6_in.W), (p7_in.H, p7_in.W)]
                pyr = self._make_pyr(p3_in.B, sizes)
                fcat = self._alloc_pyr(pyr, p3_in.C, train)
                ov = [fcat[pyr["row0"][l]:pyr["row0"][l] + pyr["rows"][l]] for l in range(5)]
                self._pyr, self._fcat = pyr, fcat
            else:
                ov = [None] * 5
            p6_up = self._node(cell, "conv6_up", "p6_w1", p6_in, None, p7_in, None, train, tape)
            p5_up = self._node(cell, "conv5_up", "p5_w1", p5_in, None, p6_up, None, train, tape)
            p4_up = self._node(cell, "conv4_up", "p4_w1", p4_in, None, p5_up, None, train, tape)
            p3_out = self._node(cell, "conv3_up", "p3_w1", p3_in, None, p4_up, None, train, tape, y=ov[0])
            if c == 0:
                p4_in = self._down_channel(f"{cell}.p4_down_channel_2", taps[1], train, tape, lazy=lz)
                p5_in = self._down_channel(f"{cell}.p5_down_channel_2", taps[2], train, tape, lazy=lz)
            p4_out = self._node(cell, "conv4_down", "p4_w2", p4_in, p4_up, None, p3_out, train, tape, y=ov[1])
            p5_out = self._node(cell, "conv5_down", "p5_w2", p5_in, p5_up, None, p4_out, train, tape, y=ov[2])
            p6_out = self._node(cell, "conv6_down", "p6_w2", p6_in, p6_up, None, p5_out, train, tape, y=ov[3])
            p7_out = self._node(cell, "conv7_down", "p7_w2", p7_in, None, None, p6_out, train, tape, y=ov[4])
            feats = [p3_out, p4_out, p5_out, p6_out, p7_out]
        return feats

    def _make_pyr(self, B: int, sizes):
        rows = [B * h * w for h, w in sizes]
        row0 = [0]
        for r in rows:
            row0.append(row0[-1] + (r + 127) // 128 * 128)
        flat = [len(sizes), B]
        for h, w in sizes:
            flat += [h, w]
        return {"desc": (ctypes.c_int * len(flat))(*flat), "row0": row0, "rows": rows, "sizes": sizes, "B": B,
                "total": row0[-1], "padded": row0[-1] != sum(rows)}

    def _alloc_pyr(self, pyr, C: int, zero_pad: bool = True) -> torch.Tensor:
        """zero_pad=False: a frozen net's tensor - nothing multiplies its padding rows into a sum (no weight gradients, no batch statistics), whatever
        they hold stays in the padding rows of its consumers' outputs (D4 / 768^2, whose 6^2 level is padded: 40 fill launches less on the pack's chain)."""
        t = self._alloc(pyr["total"], C)
        if pyr["padded"] and zero_pad:          # padding rows are multiplied in the weight-gradient GEMMs: they must be zeros, not stale bits
            for l in range(len(pyr["rows"])):       # (only the pad rows: a whole-buffer memset per pyramid tensor cost 1.8 ms/step at 768)
                lo, hi = pyr["row0"][l] + pyr["rows"][l], pyr["row0"][l + 1]
                if hi > lo:
                    call("mmd_memset_async", t[lo:hi], 0, (hi - lo) * C * 4)
        return t

    def _head(self, hname: str, feats: List[Feat], per_anchor: int, out: torch.Tensor, A: int, out_act: int,
              train: bool, tape: dict):
        """Regressor / Classifier over all 5 pyramid levels per launch (conv weights shared across levels, BN per level:
        bn_list.<lvl>.<i> sit head_layers*C channels apart in the contiguous BN arrays)."""
        ps, spec, pyr = self.ps, self.spec, self._pyr
        C, nl = feats[0].C, spec.head_layers
        nout = spec.num_anchors * per_anchor
        lev_stride = nl * C
        off0 = ps.bn_off[f"{hname}.bn_list.0.0"]
        for lvl in range(5):
            for i in range(nl):
                assert ps.bn_off[f"{hname}.bn_list.{lvl}.{i}"] == off0 + lvl * lev_stride + i * C
        desc = pyr["desc"]
        # Pack forward (grouped frozen nets): the pyramid GEMM's 128-row tiles must not straddle two nets' rows, i.e. every level needs
        # images_per_group * H * W % 128 == 0.  D4 / 768^2 at B = 8 misses that on the 6 x 6 level only (288 rows per net): the launches
        # then cover the leading levels and each trailing level runs as a plain launch of its own - few rows, so the GEMM lands on the
        # 32-row skinny kernel, whose tiles divide 288 (csrc/pw_gemm.hip pw_dispatch checks the tile height of the kernel it takes).
        nsplit = 5
        if self._grp is not None:
            good = [(self._grp[1] * h * w) % 128 == 0 for h, w in pyr["sizes"]]
            nsplit = good.index(False) if False in good else 5
            if any(good[nsplit:]) or nsplit == 0:
                raise RuntimeError("pack forward: pyramid levels %r at %d images per net do not split into whole-tile and small trailing levels" % (pyr["sizes"], self._grp[1]))
            if nsplit < 5:
                key = ("desc", nsplit)
                if key not in pyr:
                    flat = [nsplit, pyr["B"]]
                    for h, w in pyr["sizes"][:nsplit]:
                        flat += [h, w]
                    pyr[key] = (ctypes.c_int * len(flat))(*flat)
                desc = pyr[key]
        tail = list(range(nsplit, 5))

        def lv(t, l):                      # level l's rows of a pyramid row buffer
            return t[pyr["row0"][l]:pyr["row0"][l] + pyr["rows"][l]]

        def dw_tail(x, wkey, y, xf):       # the trailing levels of a depthwise pyramid launch, one plain launch each
            sc, sh, act = xf[0], xf[1], xf[2]
            for l in tail:
                h, w = pyr["sizes"][l]
                lo = l * lev_stride
                self._c("mmd_dwconv_fwd", lv(x, l), ps.w(wkey), lv(y, l), pyr["B"], h, w, C, 3, 1, sc[lo:] if sc is not None else None,
                        sh[lo:] if sh is not None else None, act, None, None, None, 0, None, None, NONE, None, None, None, 0)

        cur, cur_xf = self._fcat, (None, None, NONE, None, None, None)     # (scale, shift, act, stats, gamma, beta)
        layers = []
        for i in range(nl):
            cname = f"{hname}.conv_list.{i}"
            o = off0 + i * C
            zd = self._alloc_pyr(pyr, C, train)
            self._c("mmd_dwconv3_pyr", cur, ps.w(f"{cname}.depthwise_conv.conv.weight"), zd, desc, C, 0, *cur_xf, lev_stride,
                 None, None, None, NONE, None, None, None, None)
            dw_tail(cur, f"{cname}.depthwise_conv.conv.weight", zd, cur_xf)
            z = self._alloc_pyr(pyr, C, train)
            st = self.stats_flat[2 * o:] if train else None
            self._c("mmd_pwconv_fwd_pyr" + self._sfx, zd, ps.w(f"{cname}.pointwise_conv.conv.weight"), z, desc, C, C,
                 ps.w(f"{cname}.pointwise_conv.conv.bias"), NONE, st, lev_stride, 0, None)
            for l in tail:
                h, w = pyr["sizes"][l]
                self._c("mmd_pwconv_fwd" + self._sfx, lv(zd, l), ps.w(f"{cname}.pointwise_conv.conv.weight"), lv(z, l), pyr["rows"][l], C, C,
                        None, None, NONE, None, None, None, 0, None, h * w, ps.w(f"{cname}.pointwise_conv.conv.bias"), None, None, NONE,
                        None, None, 0, 0, None, 0)
            if train:
                for lvl in range(5):
                    ol = o + lvl * lev_stride
                    self.bn_count_host[ol:ol + C] = float(pyr["rows"][lvl])
                nxt_xf = (None, None, SWISH, st, ps.flat[ps.gamma_off + o:], ps.flat[ps.beta_off + o:])
            else:
                nxt_xf = (ps.fold_scale[o:], ps.fold_shift[o:], SWISH, None, None, None)
            layers.append({"x": cur, "x_off": None if i == 0 else off0 + (i - 1) * C, "zd": zd, "z": z, "off": o})
            cur, cur_xf = z, nxt_xf
        zd = self._alloc_pyr(pyr, C, train)
        self._c("mmd_dwconv3_pyr", cur, ps.w(f"{hname}.header.depthwise_conv.conv.weight"), zd, desc, C, 0, *cur_xf, lev_stride,
             None, None, None, NONE, None, None, None, None)
        dw_tail(cur, f"{hname}.header.depthwise_conv.conv.weight", zd, cur_xf)
        aoff, yoff = 0, []
        for (h, w) in pyr["sizes"]:
            yoff.append(aoff * per_anchor)
            aoff += h * w * spec.num_anchors
        yoff_c = (ctypes.c_longlong * 5)(*yoff)
        self._c("mmd_pwconv_fwd_pyr" + self._sfx, zd, ps.w(f"{hname}.header.pointwise_conv.conv.weight"), out, desc, C, nout,
             ps.w(f"{hname}.header.pointwise_conv.conv.bias"), out_act, None, 0, A * per_anchor, yoff_c)
        for l in tail:
            h, w = pyr["sizes"][l]
            self._c("mmd_pwconv_fwd" + self._sfx, lv(zd, l), ps.w(f"{hname}.header.pointwise_conv.conv.weight"), out, pyr["rows"][l], C, nout,
                    None, None, NONE, None, None, None, 0, None, h * w, ps.w(f"{hname}.header.pointwise_conv.conv.bias"), None, None, out_act,
                    None, None, A * per_anchor, yoff[l], None, 0)
        if train:
            tape[hname] = {"layers": layers, "hx": cur, "hx_off": off0 + (nl - 1) * C, "hzd": zd, "yoff": yoff,
                           "lev_stride": lev_stride}

    # ------------------------------------------------------------------ backward (student)
    def _slot(self, f: Feat) -> GradSlot:
        k = f.z.data_ptr()
        s = self._bw["slots"].get(k)
        if s is None:
            s = self._bw["slots"][k] = GradSlot(self._uses.get(k, 0))
        return s

    def _contrib(self, f: Feat, can_sum: bool):
        """One gradient contribution to `f` is about to be written.  -> (slot, xs); xs = (z, mean, invstd, sums, mul_b, rows_per_image)
        when this is the LAST contribution, f is the output of a BatchNorm and the writing kernel can take that BatchNorm's backward sums
        over the completed gradient (`can_sum`); the caller passes xs to the kernel, which marks the slot's sums valid."""
        s = self._slot(f)
        s.remaining -= 1
        info = self._bnout.get(f.z.data_ptr())
        if FOLD_SUMS and POOL_SCATTER and info is not None and self.ps.flat.is_cuda and f.z.data_ptr() in self._linear:
            # linear mode (the tensor is max-pooled by some node, whose backward scatters into its gradient): every contribution adds the
            # sums of its OWN share (7th element = 1); they are complete when the last one has, provided none was unable to
            # Exclusivity: the node backward's LDS-tile scatter reads the interior of this gradient at block start and stores it
            # plainly at block end, so a contribution running BESIDE it on another stream would be lost without a trace (global atomics used
            # to make that safe).  Every contribution to a scattered gradient therefore has to be issued on ONE stream.
            cur = torch.cuda.current_stream().cuda_stream
            if s.stream is None:
                s.stream = cur
            elif s.stream != cur:
                raise RuntimeError("contributions to a scattered (max-pooled) gradient issued on two streams: mmd_bifpn_node_bwd_full's "
                                   "LDS-tile scatter needs exclusive access to it (include/mmdistill.h, EXCLUSIVITY)")
            if not can_sum:
                s.linear_ok = False
                return s, None
            z, mu, istd, C, mul_b, rpi = info
            if s.sums is None:
                s.sums = self._zalloc((2 * C,), torch.float64)
            s.have_sums = s.remaining == 0 and s.linear_ok
            return s, (z, mu, istd, s.sums, mul_b, rpi, 1)
        if not (FOLD_SUMS and can_sum and s.remaining == 0 and info is not None and self.ps.flat.is_cuda):
            return s, None
        z, mu, istd, C, mul_b, rpi = info
        s.sums = self._zalloc((2 * C,), torch.float64)
        s.have_sums = True
        return s, (z, mu, istd, s.sums, mul_b, rpi, 0)

    def _acc(self, f: Feat, src: torch.Tensor):
        """gradient of f (+)= src ; the first writer just adopts the tensor (a contribution without BatchNorm sums)."""
        slot, _ = self._contrib(f, False)
        if slot.t is None:
            slot.t = src
        else:
            call("mmd_scale_acc", src, slot.t, None, 0, 0, 1, src.numel())

    @contextlib.contextmanager
    def _wgrad_stream(self):
        """Weight-gradient kernels are leaves of the backward graph (only the optimizer reads them), so they are
        issued on their own stream - a parallel branch of the captured graph - ordered after everything the
        current stream has enqueued so far; backward() joins the stream at its end.  Arena buffers are never
        reused within a step, so the only ordering needed is producer -> wgrad.
        (Round 4: the small leaves - depthwise / squeeze-excite / bias gradients - on a stream of their own, so that they do not queue behind a
        grouped flush's ~0.9 ms kernel, measured WORSE, all of them or only the last few: 14.72 - 14.89 -> 14.88 - 15.02 ms/step - one more
        branch beside the main chain costs more than the 0.1 ms the last two leaves wait.)"""
        if not self.ps.flat.is_cuda or os.environ.get("MMD_NO_WG"):
            yield
            return
        if self._wg is None:
            self._wg = torch.cuda.Stream()
        self._wg.wait_event(torch.cuda.current_stream().record_event())
        with torch.cuda.stream(self._wg):
            if _lib.dev_switch("MMD_DEV_SKIP_WG"):      # timing experiment only (gradients are WRONG): what do the weight-gradient
                global call                            # kernels cost the step through contention with the main chain?
                saved, call = call, (lambda *a, **k: 0)
                try:
                    yield
                finally:
                    call = saved
                return
            yield

    def _pw_wgrad(self, dy: torch.Tensor, xz: torch.Tensor, dw: torch.Tensor, M: int, K: int, N: int, in_scale=None, in_shift=None,
                  in_act: int = NONE, gate=None, rpi: int = 1):
        """dW[N,K] (+)= dY^T pro(X).  Grouped mode: recorded, computed by _wg_flush() at the end of the backward segment."""
        # precision "bf16": grouped as well (round 4, mmd_wgrad_grouped_bf16) where the per-layer launches are skeleton-bound - D2 / 512^2 in
        # bf16 15.50 -> 14.64 ms/step; at D4 / 768^2 the layers are tall enough that the per-layer launches on the side stream measure the
        # same or better (53.0-53.3 vs 53.5-53.8 ms), so tall nets keep them (rows of the stem output decide; MMD_WG_GROUP_BF16=0/1 forces)
        if not WG_GROUP or not self.ps.flat.is_cuda or (self.precision != "fp32" and not self._group_bf16()):
            with self._wgrad_stream():
                call("mmd_pwconv_bwd_weight" + self._sfx, dy, xz, dw, M, K, N, in_scale, in_shift, in_act, gate, rpi)
            return
        self._wg_pending.append((dy, xz, dw, M, K, N, in_scale, in_shift, in_act, gate, rpi))
        self._wg_count += 1
        if WG_POINTS:
            if self._wg_count in WG_POINTS:
                self._wg_flush(final=False)
        elif WG_CHUNK > 0 and self._wg_count % WG_CHUNK == 0:
            # (counted over the whole backward, not per segment: the split backward of the data-parallel step then flushes at the same layers as
            # the one-segment form - its second segment's 11 layers would otherwise all sit in the exposed last flush)
            self._wg_flush(final=False)

    def _group_bf16(self) -> bool:
        env = os.environ.get("MMD_WG_GROUP_BF16")
        if env is not None:
            return env not in ("", "0")
        stem = self.tape.get("stem")
        return stem is None or stem[1].M <= 600000

    def _leaf(self, fn):
        """A leaf of the backward graph other than a 1x1-conv weight gradient (depthwise / squeeze-excite / fusion-weight / bias
        gradients): in grouped mode it is issued by _wg_flush() with the rest, behind ONE cross-stream dependency, instead of forking
        off the main chain at ~160 points of the step (every fork is an event record + wait = a cross-branch edge of the graph)."""
        if WG_DEFER and self.ps.flat.is_cuda:
            self._leaf_pending.append(fn)
        else:
            with self._wgrad_stream():
                fn()

    def _wg_flush(self, final: bool = True):
        """Launch the deferred weight gradients of this backward segment: one persistent grid over all (layer, tile, split) items
        and one fold, on the weight-gradient stream behind everything issued so far.  final: nothing of the main chain runs beside
        this flush (end of a backward segment): full-width grid; otherwise the thin one (WG_BLOCKS_MID)."""
        pend, self._wg_pending = self._wg_pending, []
        leaves, self._leaf_pending = self._leaf_pending, []
        seg = self._wg_segment
        self._wg_segment += 1
        if leaves:
            with self._wgrad_stream():
                for fn in leaves:
                    fn()
        if not pend:
            return
        ptr = lambda t: 0 if t is None else t.data_ptr()
        sig = tuple((ptr(d), ptr(x), ptr(w), M, K, N, ptr(a), ptr(b), act, ptr(gt), rpi) for d, x, w, M, K, N, a, b, act, gt, rpi in pend)
        # one plan per (segment, operand signature): a second step variant captured later ("aug") may lay its operands out
        # differently, and the device tables of the variants captured before must stay alive - their graphs keep reading them
        plan = self._wg_plans.get((seg, sig))
        if plan is None:
            if self.arena.frozen:
                raise RuntimeError("weight-gradient operands moved after graph capture")
            n = len(pend)
            arr = (WgLayer * n)()
            for i, (d, x, w, M, K, N, a, b, act, gt, rpi) in enumerate(pend):
                arr[i].dy, arr[i].x, arr[i].dw = ptr(d), ptr(x), ptr(w)
                arr[i].in_scale, arr[i].in_shift, arr[i].gate = ptr(a) or None, ptr(b) or None, ptr(gt) or None
                arr[i].M, arr[i].K, arr[i].N, arr[i].in_act, arr[i].rows_per_image = M, K, N, act, max(int(rpi), 1)
            ni, nt, wsf = ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong()
            rc = _lib.LIB.load().mmd_wgrad_plan(ctypes.cast(arr, ctypes.c_void_p), n, WG_ROWS_FINAL if final else WG_ROWS, ctypes.cast(ctypes.pointer(ni), ctypes.c_void_p),
                                                ctypes.cast(ctypes.pointer(nt), ctypes.c_void_p), ctypes.cast(ctypes.pointer(wsf), ctypes.c_void_p))
            if rc != 0:
                raise RuntimeError(f"mmd_wgrad_plan failed with status {rc}")
            table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
            plan = {"sig": sig, "table": table, "n": n, "items": ni.value, "tiles": nt.value, "ws": wsf.value,
                    "flops": float(sum(2.0 * M * K * N for _, _, _, M, K, N, *_ in pend)),
                    "bytes": float(sum(4.0 * (M * K + M * N + N * K) for _, _, _, M, K, N, *_ in pend)),
                    # every layer's M a multiple of 32 (any net at B >= 2): the launch runs without row clamps / masks (mmd_wgrad_grouped_form)
                    "rows32": int(all(M % 32 == 0 for _, _, _, M, *_ in pend) and not os.environ.get("MMD_NO_WG_R32"))}
            self._wg_plans[(seg, sig)] = plan
        ws = self._alloc(plan["ws"])
        with self._wgrad_stream():
            call("mmd_wgrad_grouped_form", plan["table"], plan["n"], plan["items"], plan["tiles"], ws, WG_BLOCKS if final else WG_BLOCKS_MID,
                 plan["flops"], plan["bytes"], 1 if self._sfx else 0, plan["rows32"])

    def _bn_bwd(self, g_in: torch.Tensor, z: torch.Tensor, aff, bn_name: str, act: int, M: int, C: int, rpi: int = 0,
                mul_bc=None, mul_b=None, add_bc=None, sums=None, lazy: bool = False):
        """BN(+act) backward: returns dz (gradient w.r.t. the conv's raw output); dgamma/dbeta accumulated.
        `sums` given = the per-channel sums were produced elsewhere (fused into another pass): pass 2 only.
        lazy: no pass 2 at all - returns a LazyDz that the 1x1 conv's two gradient GEMMs evaluate in their prologues."""
        b = self.ps.bn(bn_name)
        if sums is None:
            sums = self._zalloc((2 * C,), torch.float64)
            call("mmd_bn_bwd_reduce", g_in, z, aff[0], aff[1], aff[2], aff[3], act, mul_bc, mul_b, add_bc, rpi, None, sums, M, C,
                 *self._stats_ws(sums, M, C))
        if lazy and LAZY_BN and mul_bc is None and add_bc is None:
            return LazyDz(g_in, z, aff, sums, act, mul_b, rpi, bn_name, M)
        dz = self._alloc(M, C)
        call("mmd_bn_bwd_apply", g_in, z, aff[2], aff[3], b["gamma"], sums, M, dz, b["dgamma"], b["dbeta"], M, C,
             aff[0], aff[1], act, mul_bc, mul_b, add_bc, rpi)
        return dz

    def _pw_bwd(self, dz: torch.Tensor, x: Feat, wkey: str, N: int, bias_key: Optional[str], want_dx: bool,
                gate=None, plain_in=False, into: Optional[Feat] = None, pool5: Optional[tuple] = None) -> Optional[torch.Tensor]:
        """Gradients of a 1x1 conv.  into = the Feat whose gradient slot receives dx (normally x itself): dx is then written into / on top
        of that slot by the GEMM's epilogue (no scale_acc launch), together with the backward sums of the BatchNorm that produced the
        Feat when this is the slot's last contribution; returns None in that case."""
        ps = self.ps
        M, K = x.M, x.C
        if isinstance(dz, LazyDz):
            # BatchNorm backward in the operand prologue of the input-gradient GEMM (a bias in front of a BatchNorm has an exactly-
            # zero gradient).  That launch also stores the evaluated dz once and adds dgamma / dbeta, so the weight gradient - a
            # leaf on the side stream, ordered behind it - is the plain GEMM on one tensor
            L, b = dz, ps.bn(dz.bn_name)
            bnargs = (L.aff[0], L.aff[1], L.aff[2], L.aff[3], L.sums, L.count, L.act, L.mul_b, L.rpi)
            xargs = (None if plain_in else x.scale, None if plain_in else x.shift, NONE if plain_in else x.act, gate, x.H * x.W)
            self._wg_read_done = None
            if not want_dx:
                with self._wgrad_stream():
                    call("mmd_pwconv_bwd_weight_bn" + self._sfx, L.g, L.z, x.z, ps.g(wkey), M, K, N, *xargs, *bnargs, b["dgamma"],
                         b["dbeta"])
                    self._wg_read_done = torch.cuda.current_stream().record_event() if ps.flat.is_cuda else None
                return None
            dzm = self._alloc(M, N)
            if (into is not None and FOLD_SUMS and MBW_FUSED and M >= MBW_MIN_ROWS and L.act == SWISH and L.mul_b is None and gate is None
                    and x.scale is None and x.bn is None and x.act == NONE and ps.flat.is_cuda
                    and _lib.LIB.load().mmd_mbconv_expand_bwd_supported(K, N) == 1):
                # thin-input expand conv of a high-resolution block: one pass over (g0, z0) - input gradient, weight gradient, no dz0 in HBM
                slot, xs = self._contrib(into, True)
                assert xs is None or not xs[6], "the epilogue sums the accumulated total: not a linear-sum (pooled) tensor"
                residual = slot.t
                if slot.t is None:
                    slot.t = self._alloc(M, K)
                xsa = (None, None, None, None, 0, None) if xs is None else (xs[0], xs[1], xs[2], xs[4], xs[5], xs[3])
                call("mmd_mbconv_expand_bwd_fused", L.g, L.z, x.z, ps.w(wkey), slot.t, residual, ps.g(wkey), M, K, N, L.aff[0], L.aff[1], L.aff[2],
                     L.aff[3], L.sums, L.count, b["dgamma"], b["dbeta"], *xsa)
                return None
            if into is not None and FOLD_SUMS:
                slot, xs = self._contrib(into, True)
                assert xs is None or not xs[6], "the GEMM epilogue sums the accumulated total: not a linear-sum (pooled) tensor"
                residual = slot.t
                if slot.t is None:
                    slot.t = self._alloc(M, K)
                xsargs = (None, None, None, None, 0, None, None, 0)
                if xs is not None:
                    xsargs = (xs[0], xs[1], xs[2], xs[4], xs[5], xs[3], *self._stats_ws(xs[3], M, K))
                if self._sfx:
                    call("mmd_pwconv_bwd_data_bn2" + self._sfx, L.g, L.z, ps.w_t(wkey), slot.t, M, K, N, *bnargs, dzm, b["dgamma"], b["dbeta"],
                         residual, *xsargs, None, None, None, None, None, None, 0)
                else:
                    # (the GEMM reduces over N = the conv's output channels: the slab kernel's "K")
                    call("mmd_pwconv_bwd_data_bn2_form", L.g, L.z, ps.w_t(wkey), slot.t, M, K, N, *bnargs, dzm, b["dgamma"], b["dbeta"],
                         residual, *xsargs, None, None, None, None, None, None, 0, *self._slab_ws(M, N, K, 1), 0)
                self._pw_wgrad(dzm, x.z, ps.g(wkey), M, K, N, *xargs)
                return None
            dx = self._alloc(M, K)
            if pool5 is not None:
                # MBConv project conv: the pooled pass of the squeeze-excite / BatchNorm-1 backward over (z1, dx) rides in this launch's epilogue
                call("mmd_pwconv_bwd_data_bn2" + self._sfx, L.g, L.z, ps.w_t(wkey), dx, M, K, N, *bnargs, dzm, b["dgamma"], b["dbeta"],
                     None, None, None, None, None, 0, None, None, 0, *pool5)
                self._pw_wgrad(dzm, x.z, ps.g(wkey), M, K, N, *xargs)
                return dx
            call("mmd_pwconv_bwd_data_bn" + self._sfx, L.g, L.z, ps.w_t(wkey), dx, M, K, N, *bnargs, dzm, b["dgamma"], b["dbeta"])
            self._pw_wgrad(dzm, x.z, ps.g(wkey), M, K, N, *xargs)
            if into is not None:
                self._acc(into, dx)
                return None
            return dx
        if bias_key:
            self._leaf(lambda dz=dz, gb=ps.g(bias_key): call("mmd_colsum", dz, gb, M, N))
        self._pw_wgrad(dz, x.z, ps.g(wkey), M, K, N, None if plain_in else x.scale, None if plain_in else x.shift,
                       NONE if plain_in else x.act, gate, x.H * x.W)
        if not want_dx:
            return None
        dx = self._alloc(M, K)
        call("mmd_pwconv_bwd_data" + self._sfx, dz, ps.w_t(wkey), dx, M, K, N, 0)
        if into is not None:
            self._acc(into, dx)
            return None
        return dx

    def _dw_bwd(self, dzd: torch.Tensor, x: Feat, wkey: str, k: int, s: int, want_dx: bool = True, bn_aff=None):
        """-> dx, or (dx, sums) when `bn_aff` = (scale, shift, mean, invstd) of the BatchNorm(+swish) that produced x: the
        stride-1 input-gradient launch then also accumulates that BN's backward sums (no separate reduce pass) and the conv's weight
        gradient (tile kernel; for the 3x3 layers that gives up the faster row-streaming input-gradient kernel and still wins: a
        weight-gradient leaf costs its full kernel time on the saturated chip - 18.44 -> 18.27 ms/step)."""
        ps = self.ps
        wg_inside = self.DW_WG and want_dx and bn_aff is not None and ((s == 1 and x.C >= 64) or (s == 2 and DW_S2_SUMS and DW_S2_WG))      # (block 0, 32 channels at 256^2: 221 us inside vs 47 + 67)
        if not wg_inside:
            self._leaf(lambda x=x, dzd=dzd, gw=ps.g(wkey): call("mmd_dwconv_bwd_weight", x.z, dzd, gw, x.B, x.H, x.W, x.C, k, s, x.scale, x.shift,
                                                                 x.act))
        if not want_dx:
            return None
        dx = self._alloc(x.M, x.C)
        if bn_aff is not None and (s == 1 or DW_S2_SUMS):      # (stride 2, round 5: the sums ride in the gather launch too - no reduce pass)
            sums = self._zalloc((2 * x.C,), torch.float64)
            call("mmd_dwconv_bwd_data", dzd, ps.w(wkey), dx, x.B, x.H, x.W, x.C, k, s, x.z, bn_aff[0], bn_aff[1], bn_aff[2],
                 bn_aff[3], sums, *self._stats_ws(sums, x.M, x.C), ps.g(wkey) if wg_inside else None)
            return dx, sums
        call("mmd_dwconv_bwd_data", dzd, ps.w(wkey), dx, x.B, x.H, x.W, x.C, k, s, None, None, None, None, None, None, None, 0, None)
        return (dx, None) if bn_aff is not None else dx

    PYR_WG = not os.environ.get("MMD_NO_PYR_WG")

    def _pyr_dw_bwd(self, dzd, wkey: str, g, desc, C: int, ls: int, x, sc, sh, act: int, bn=None) -> bool:
        """Input gradient of a shared-weight pyramid depthwise conv; its weight gradient rides in the same launch, and so do the sums of
        the BatchNorm(+swish) backward that consumes g when `bn` = (mean, invstd, sums) of x's producer is given (-> True: the caller
        skips that reduce pass).  MMD_NO_PYR_WG=1: weight gradient by its own launch on the weight-gradient stream (earlier form, A/B)."""
        ps = self.ps
        if self.PYR_WG:
            mu, istd, sums = bn if bn is not None else (None, None, None)
            call("mmd_dwconv3_pyr", dzd, ps.w(wkey), g, desc, C, 1, None, None, NONE, None, None, None, ls, x, sc, sh, act, ps.g(wkey),
                 mu, istd, sums)
            return bn is not None
        self._leaf(lambda: call("mmd_dwconv3_pyr_bwd_weight", x, dzd, ps.g(wkey), desc, C, sc, sh, act, ls))
        call("mmd_dwconv3_pyr", dzd, ps.w(wkey), g, desc, C, 1, None, None, NONE, None, None, None, 0, None, None, None, NONE, None,
             None, None, None)
        return False

    def _head_bwd(self, hname: str, per_anchor: int, dout: torch.Tensor, tape: dict, pyr: dict, A: int, C: int):
        """Backward of one head over the whole pyramid; returns the gradient w.r.t. the pyramid feature buffer."""
        spec, ps = self.spec, self.ps
        desc, Mt = pyr["desc"], pyr["total"]
        rec = tape[hname]
        nout = spec.num_anchors * per_anchor
        ls = rec["lev_stride"]
        dy = self._alloc_pyr(pyr, nout, False)          # (the slice launch writes the padding rows itself; the plain input-gradient GEMMs below write every row)
        call("mmd_slice_rows_pyr", dout, dy, desc, nout, A * per_anchor, (ctypes.c_longlong * len(rec["yoff"]))(*[int(v) for v in rec["yoff"]]))
        hw_key = f"{hname}.header.pointwise_conv.conv.weight"
        self._leaf(lambda dy=dy, gb=ps.g(f"{hname}.header.pointwise_conv.conv.bias"): call("mmd_colsum", dy, gb, Mt, nout))
        self._pw_wgrad(dy, rec["hzd"], ps.g(hw_key), Mt, C, nout)
        dzd = self._alloc_pyr(pyr, C, False)
        call("mmd_pwconv_bwd_data" + self._sfx, dy, ps.w_t(hw_key), dzd, Mt, C, nout, 0)
        xo = rec["hx_off"]
        # the depthwise weight gradient rides in the (flipped) input-gradient launch: a leaf launch's kernel time is paid in full; so do
        # the sums of the BatchNorm backward that consumes g (x's producer), instead of a reduce pass over (g, z)
        bsums = self._zalloc((2 * 5 * ls,), torch.float64)
        off0 = rec["layers"][0]["off"]
        g = self._alloc_pyr(pyr, C)
        have_sums = self._pyr_dw_bwd(dzd, f"{hname}.header.depthwise_conv.conv.weight", g, desc, C, ls, rec["hx"], self.t_scale[xo:], self.t_shift[xo:],
                                     SWISH, bn=(self.t_mean[xo:], self.t_invstd[xo:], bsums[2 * (xo - off0):]))
        for i in reversed(range(spec.head_layers)):
            L = rec["layers"][i]
            cname = f"{hname}.conv_list.{i}"
            o = L["off"]
            sums = bsums[2 * (o - off0):]
            if not have_sums:
                call("mmd_bn_bwd_reduce_pyr", g, L["z"], self.t_scale[o:], self.t_shift[o:], self.t_mean[o:], self.t_invstd[o:],
                     SWISH, desc, ls, None, sums, C)
            dz = self._alloc_pyr(pyr, C)
            call("mmd_bn_bwd_apply_pyr", g, L["z"], self.t_mean[o:], self.t_invstd[o:], ps.flat[ps.gamma_off + o:], sums,
                 desc, ls, dz, ps.grad[ps.gamma_off + o:], ps.grad[ps.beta_off + o:], C, self.t_scale[o:], self.t_shift[o:],
                 SWISH)
            wkey = f"{cname}.pointwise_conv.conv.weight"
            self._pw_wgrad(dz, L["zd"], ps.g(wkey), Mt, C, C)
            dzd = self._alloc_pyr(pyr, C, False)
            call("mmd_pwconv_bwd_data" + self._sfx, dz, ps.w_t(wkey), dzd, Mt, C, C, 0)
            xo = L["x_off"]
            g = self._alloc_pyr(pyr, C)
            have_sums = self._pyr_dw_bwd(dzd, f"{cname}.depthwise_conv.conv.weight", g, desc, C, ls, L["x"], None if xo is None else self.t_scale[xo:],
                                         None if xo is None else self.t_shift[xo:], NONE if xo is None else SWISH,
                                         bn=None if xo is None else (self.t_mean[xo:], self.t_invstd[xo:], bsums[2 * (xo - off0):]))
        return g

    def backward(self, dcls_logit: torch.Tensor, dreg: torch.Tensor, dfeats: List[Optional[torch.Tensor]],
                 stop_before: Optional[int] = None, dfeat_pyr: Optional[torch.Tensor] = None):
        """Accumulates parameter gradients into ps.grad.  dcls_logit [B,A,NC] is the gradient w.r.t. the
        classifier header's PRE-sigmoid output, dreg [B,A,4], dfeats[l] (nullable) w.r.t. the BiFPN outputs.
        stop_before=k: stop after backbone block k (heads, BiFPN and blocks >= k done, weight-gradient stream joined), so
        that the caller can start the all-reduce of those gradients; backward_finish() runs blocks < k and the stem.
        dfeat_pyr: the same gradients as dfeats as ONE pyramid row buffer (dfeats[l] are its level slices): the heads' input gradients
        and it are then added in one launch that also takes the BatchNorm-backward sums of the last BiFPN cell's five output nodes."""
        spec, ps, tape = self.spec, self.ps, self.tape
        feats: List[Feat] = tape["feats"]
        A = tape["A"]
        slots: Dict[int, GradSlot] = {}
        self._bw = {"slots": slots, "stem_sums": None}
        self._wg_pending, self._leaf_pending, self._wg_segment, self._se_wg = [], [], 0, []
        self._wg_count = 0
        slot = self._slot
        fold_heads = FOLD_SUMS and ps.flat.is_cuda and (dfeat_pyr is not None or all(d is None for d in dfeats))
        if not fold_heads:
            for f, d in zip(feats, dfeats):
                if d is not None:
                    slot(f).t = d
        # ---- heads: every layer handles the whole pyramid in one launch
        pyr = self._pyr
        desc, Mt = pyr["desc"], pyr["total"]
        C = feats[0].C
        # the two heads are independent until their input gradients meet: the regressor runs on a side stream
        # (a parallel branch of the captured graph); bsums / workspaces are allocated here on the host, in order
        main_stream = torch.cuda.current_stream() if ps.flat.is_cuda else None
        if main_stream is not None and self._side is None:
            self._side = torch.cuda.Stream()
        fork = main_stream.record_event() if main_stream is not None else None
        gparts = []
        for hname, per_anchor, dout in (("classifier", spec.num_classes, dcls_logit), ("regressor", 4, dreg)):
            use_side = main_stream is not None and hname == "regressor" and not os.environ.get("MMD_NO_SIDE")
            if use_side:
                self._side.wait_event(fork)
            with torch.cuda.stream(self._side if use_side else main_stream):
                gparts.append(self._head_bwd(hname, per_anchor, dout, tape, pyr, A, C))
        if main_stream is not None:
            main_stream.wait_stream(self._side)
        gsum = gparts[0]
        if fold_heads:
            # classifier + regressor (+ MTA) gradients in one launch, which also takes the BatchNorm-backward sums of the five output nodes
            vp = ctypes.c_void_p
            ptrs = [[None] * 5 for _ in range(4)]
            for lvl, f in enumerate(feats):
                sl, xs = self._contrib(f, True)
                sl.t = gsum[pyr["row0"][lvl]:pyr["row0"][lvl] + pyr["rows"][lvl]]
                if xs is not None:
                    for k in range(4):
                        ptrs[k][lvl] = xs[k].data_ptr()
            arrs = [(vp * 5)(*p_) for p_ in ptrs]
            call("mmd_pyr_add_bnsums", gparts[0], gparts[1], dfeat_pyr, gsum, desc, C, *arrs)
        else:
            call("mmd_scale_acc", gparts[1], gsum, None, 0, 0, 1, gsum.numel())
            for lvl, f in enumerate(feats):
                self._acc(f, gsum[pyr["row0"][lvl]:pyr["row0"][lvl] + pyr["rows"][lvl]])
        # ---- BiFPN (cells and nodes in reverse)
        n_nodes = sum(len(tape[f"bifpn.{c}.nodes"]) for c in range(spec.fpn_cells))
        wdot_all = self._zalloc((4 * n_nodes,))
        theta_desc, node_i = [], 0
        xsn = lambda xs: (None, None, None, None) if xs is None else xs[:4]
        for c in reversed(range(spec.fpn_cells)):
            cell = f"bifpn.{c}"
            for rec in reversed(tape[cell + ".nodes"]):
                out: Feat = rec["out"]
                s = slot(out)
                if s.t is None:
                    continue          # node output unused downstream (cannot happen in this topology)
                name = f"{cell}.{rec['conv']}"
                W = out.C
                dz = self._bn_bwd(s.t, rec["z"], rec["bn"], f"{name}.bn", NONE, out.M, W, lazy=True,
                                  sums=s.sums if s.have_sums else None)
                in0, in1, up, pl = rec["in0"], rec["in1"], rec["up"], rec["pl"]
                mode_ = (1 if in1 is not None else 0) | (2 if up is not None else 0) | (4 if pl is not None else 0)
                full = (NODE_BWD_FULL and isinstance(dz, LazyDz) and ps.flat.is_cuda and self.NODE_WG and W % 16 == 0 and W <= 224 and mode_ in (2, 5, 4)
                        and (self.precision == "fp32" or (W <= 160 and NODE_FULL_BF16)) and (pl is None or (POOL_SCATTER and FOLD_SUMS)))
                dzd = None if full else self._pw_bwd(dz, rec["zd"], f"{name}.pointwise_conv.conv.weight", W, None, True)
                # the depthwise weight gradient and the depthwise input gradient both come out of the fusion-backward launch below
                if not self.NODE_WG:
                    self._dw_bwd(dzd, rec["f"], f"{name}.depthwise_conv.conv.weight", 3, 1, want_dx=False)
                th = ps.w(f"{cell}.{rec['theta']}")
                nth = th.numel()
                # gradients of the same-resolution operands (in0, in1) come straight out of the fuse backward launch; dx is
                # only materialised for an upsampled / pooled operand
                same, xsargs, own = [], [], 0
                for oi, operand in enumerate((in0, in1)):
                    if operand is None:
                        same += [None, 0]
                        xsargs += [None] * 4
                        continue
                    sl, xs = self._contrib(operand, True)
                    accumulate = 0 if sl.t is None else 1
                    if sl.t is None:
                        sl.t = self._alloc(operand.M, W)
                    same += [sl.t, accumulate]
                    xsargs += xsn(xs)
                    own |= (1 << oi) if (xs is not None and xs[6]) else 0
                assert in1 is None or same[0].data_ptr() != same[2].data_ptr()
                # the gradient of an upsampled operand leaves the same launch (2x2 block sums); dx is only materialised for a
                # pooled operand (its scatter follows the arg-max of overlapping windows)
                scatter = POOL_SCATTER and FOLD_SUMS and pl is not None and ps.flat.is_cuda
                dx = self._alloc(out.M, W) if (pl is not None and not scatter) else None
                upargs = (None, 0)
                if up is not None:
                    su, xs = self._contrib(up, True)
                    upargs = (su.t, 1) if su.t is not None else (self._alloc(up.M, W), 0)
                    su.t = upargs[0]
                    xsargs += xsn(xs)
                    own |= 4 if (xs is not None and xs[6]) else 0
                else:
                    xsargs += [None] * 4
                # the fusion-weight gradients of all nodes are finished by ONE launch after the BiFPN (their dot products land in wdot_all)
                wdot = wdot_all[4 * node_i:4 * node_i + 4]
                theta_desc.append((ps.entries[f"{cell}.{rec['theta']}"].off, nth))
                node_i += 1
                lzb = [o is not None and o.bn is not None for o in (in0, in1, up, pl)]
                if full:
                    # whole-node backward: BatchNorm backward of the node + its 1x1 conv's input gradient + depthwise / fusion backward in ONE
                    # launch; the evaluated dz is stored once for the conv's weight-gradient GEMM (grouped flush)
                    dplargs = (None, None, None, None, None)
                    if scatter:
                        sl, xs = self._contrib(pl, True)
                        if sl.t is None:
                            sl.t = self._zalloc((pl.M, W))
                        dplargs = (sl.t, *xsn(xs))
                    vp = ctypes.c_void_p
                    ops_ = (in0, in1, up, pl)
                    scs = (vp * 4)(*[(o.scale.data_ptr() if l else None) for o, l in zip(ops_, lzb)])
                    shs = (vp * 4)(*[(o.shift.data_ptr() if l else None) for o, l in zip(ops_, lzb)])
                    L, bq = dz, ps.bn(dz.bn_name)
                    wkey = f"{name}.pointwise_conv.conv.weight"
                    dzm = self._alloc(out.M, W)
                    call("mmd_bifpn_node_bwd_full", in0.z, in1.z if in1 else None, up.z if up else None, pl.z if pl else None, th,
                         ps.w(f"{name}.depthwise_conv.conv.weight"), wdot, in0.B, in0.H, in0.W, W, *same, *upargs,
                         ps.g(f"{name}.depthwise_conv.conv.weight"), *xsargs, *dplargs, own, scs if any(lzb) else None, shs if any(lzb) else None,
                         L.g, L.z, L.aff[0], L.aff[2], L.aff[3], L.sums, L.count, ps.w_t(wkey), dzm, bq["dgamma"], bq["dbeta"])
                    zdf = rec["zd"]
                    self._pw_wgrad(dzm, zdf.z, ps.g(wkey), out.M, W, W, None, None, NONE, None, zdf.H * zdf.W)
                    continue
                if any(lzb):
                    # lazy operands: their values are read as z * scale + shift (finalized coefficients) inside the launch
                    assert scatter or pl is None
                    dplargs = (None, None, None, None, None)
                    if scatter:
                        sl, xs = self._contrib(pl, True)
                        if sl.t is None:
                            sl.t = self._zalloc((pl.M, W))
                        dplargs = (sl.t, *xsn(xs))
                    vp = ctypes.c_void_p
                    ops_ = (in0, in1, up, pl)
                    scs = (vp * 4)(*[(o.scale.data_ptr() if l else None) for o, l in zip(ops_, lzb)])
                    shs = (vp * 4)(*[(o.shift.data_ptr() if l else None) for o, l in zip(ops_, lzb)])
                    call("mmd_bifpn_node_dw_bwd3_lz", in0.z, in1.z if in1 else None, up.z if up else None, pl.z if pl else None, th,
                         ps.w(f"{name}.depthwise_conv.conv.weight"), dzd, None, wdot, in0.B, in0.H, in0.W, W, *same, *upargs,
                         ps.g(f"{name}.depthwise_conv.conv.weight") if self.NODE_WG else None, *xsargs, *dplargs, own, scs, shs)
                    continue
                if scatter or own:
                    # scatter: the pooled operand's gradient leaves this launch too - added to each window's arg-max with atomics, on top of the
                    # earlier contributions (or of zeros: a zero-initialised arena buffer when this is the first one); own: an operand that is
                    # max-pooled by ANOTHER node keeps its BatchNorm sums linearly, this launch adds the sums of its share
                    dplargs = (None, None, None, None, None)
                    if scatter:
                        sl, xs = self._contrib(pl, True)
                        if sl.t is None:
                            sl.t = self._zalloc((pl.M, W))
                        dplargs = (sl.t, *xsn(xs))
                    call("mmd_bifpn_node_dw_bwd3", in0.z, in1.z if in1 else None, up.z if up else None, pl.z if pl else None, th,
                         ps.w(f"{name}.depthwise_conv.conv.weight"), dzd, dx, wdot, in0.B, in0.H, in0.W, W, *same, *upargs,
                         ps.g(f"{name}.depthwise_conv.conv.weight") if self.NODE_WG else None, *xsargs, *dplargs, own)
                    if scatter or pl is None:
                        continue
                else:
                    call("mmd_bifpn_node_dw_bwd2", in0.z, in1.z if in1 else None, up.z if up else None, pl.z if pl else None, th,
                         ps.w(f"{name}.depthwise_conv.conv.weight"), dzd, dx, wdot, in0.B, in0.H, in0.W, W, *same, *upargs,
                         ps.g(f"{name}.depthwise_conv.conv.weight") if self.NODE_WG else None, *xsargs)
                if pl is not None:      # the pooled operand's gradient: gather over the windows whose arg-max it is
                    wi = 1 + (1 if in1 is not None else 0) + (1 if up is not None else 0)
                    sl, xs = self._contrib(pl, _lib.LIB.load().mmd_maxpool_bwd_sums_ok(pl.B, pl.H, pl.W, W) == 1)
                    accumulate = 0 if sl.t is None else 1
                    if sl.t is None:
                        sl.t = self._alloc(pl.M, W)
                    call("mmd_maxpool_same_bwd_acc2", pl.z, dx, sl.t, th, nth, wi, accumulate, pl.B, pl.H, pl.W, W, *xsn(xs))
            if c == 0:
                first = tape[cell + ".first"]
                c6, p6_in, p7_in = first["c6"], first["p6_in"], first["p7_in"]
                s7, s6 = slot(p7_in), slot(p6_in)
                if s7.t is not None:
                    self._contrib(p6_in, False)
                    acc = 0 if s6.t is None else 1
                    if s6.t is None:
                        s6.t = self._alloc(p6_in.M, p6_in.C)
                    call("mmd_maxpool_same_bwd_acc", p6_in.z, s7.t, s6.t, None, 0, 0, acc, p6_in.B, p6_in.H, p6_in.W, p6_in.C)
                if s6.t is not None:
                    sc6, xs = self._contrib(c6, True)
                    sc6.t = self._alloc(c6.M, c6.C)
                    call("mmd_maxpool_same_bwd_acc2", c6.z, s6.t, sc6.t, None, 0, 0, 0, c6.B, c6.H, c6.W, c6.C, *xsn(xs))
                for nm in ("p5_down_channel_2", "p4_down_channel_2", "p5_down_channel", "p4_down_channel",
                           "p3_down_channel", "p5_to_p6"):
                    name = f"{cell}.{nm}"
                    rec = tape[name]
                    out = rec["out"]
                    s = slot(out)
                    if s.t is None:
                        continue
                    x: Feat = rec["x"]
                    dz = self._bn_bwd(s.t, rec["z"], rec["bn"], f"{name}.1", NONE, out.M, out.C, lazy=True,
                                      sums=s.sums if s.have_sums else None)
                    self._pw_bwd(dz, x, f"{name}.0.conv.weight", out.C, None, True, into=x)
        if theta_desc:
            key = tuple(theta_desc)
            if key not in self._theta_desc:
                if self.arena.frozen:       # (under capture the upload would be recorded as a copy from a host temporary)
                    raise RuntimeError("fusion-weight gradient operands moved after graph capture")
                self._theta_desc[key] = torch.tensor(theta_desc, dtype=torch.int64, device=self.device)
            self._leaf(lambda d=self._theta_desc[key], n=len(theta_desc): call("mmd_bifpn_theta_bwd_batched", ps.flat, ps.grad, wdot_all, d, n))
        self._backward_blocks([b for b in spec.blocks if stop_before is None or b.idx >= stop_before])
        if stop_before is None:
            self.backward_finish(0)
        else:
            self._wg_flush()
            if self._wg is not None:
                torch.cuda.current_stream().wait_stream(self._wg)

    def backward_finish(self, stop_before: int):
        """Second part of a split backward: backbone blocks < stop_before, then the stem."""
        self._backward_blocks([b for b in self.spec.blocks if b.idx < stop_before])
        self._backward_stem()

    def _backward_blocks(self, blocks):
        """Backbone blocks in reverse."""
        spec, ps, tape = self.spec, self.ps, self.tape
        slot = self._slot
        P = "backbone_net.model"
        stem_sums = self._bw["stem_sums"]
        for blk in reversed(blocks):
            q = f"{P}._blocks.{blk.idx}"
            rec = tape[f"blk{blk.idx}"]
            out: Feat = rec["out"]
            inp: Feat = rec["inp"]
            s = slot(out)
            if s.t is None:
                continue              # blocks after the last tap do not exist; defensive
            dy = s.t
            f1: Feat = rec["f1"]
            M1, HW1 = f1.M, f1.H * f1.W
            dz2 = self._bn_bwd(dy, rec["z2"], rec["bn2"], f"{q}._bn2", NONE, M1, blk.cout, rpi=HW1, mul_b=rec["rs"], lazy=True,
                               sums=s.sums if s.have_sums else None)
            # squeeze-excite backward.  One pass over (z1, g1) pools d(gate) AND the partials of the BN-1 backward sums (the SE kernels finish
            # those sums once dpooled is known, so the expanded tensor is not read by a BN reduce pass); that pass rides in the epilogue of
            # the project conv's input-gradient GEMM, which produces g1
            a1 = rec["bn1"]
            pool5 = self._zalloc((5, f1.B, blk.cmid))
            p5 = (f1.z, a1[0], a1[1], a1[2], a1[3], pool5, f1.B) if (P5_IN_GEMM and isinstance(dz2, LazyDz) and ps.flat.is_cuda) else None
            g1 = self._pw_bwd(dz2, f1, f"{q}._project_conv.conv.weight", blk.cout, None, True, gate=rec["gate"], pool5=p5)
            # the skip branch may ADOPT dy as the gradient slot of the block input, and the block's own input gradient is later
            # accumulated into that buffer in place; with a lazy BatchNorm backward the project conv's weight-gradient GEMM (side
            # stream) still reads dy, so that later accumulation waits for it (the event is long past by then)
            dy_read = self._wg_read_done if (blk.skip and isinstance(dz2, LazyDz)) else None
            if blk.skip:
                self._acc(inp, dy)
            if p5 is None:
                call("mmd_chan_pool_bwd", f1.z, a1[0], a1[1], a1[2], a1[3], g1, pool5, f1.B, HW1, blk.cmid)
            dpe = self._alloc(f1.B, blk.cmid)
            dpr = self._alloc(f1.B, blk.se)
            dpooled = self._alloc(f1.B, blk.cmid)
            sums1 = self._zalloc((2 * blk.cmid,), torch.float64)
            segrads = (ps.g(f"{q}._se_reduce.conv.weight"), ps.g(f"{q}._se_reduce.conv.bias"), ps.g(f"{q}._se_expand.conv.weight"),
                       ps.g(f"{q}._se_expand.conv.bias"))
            if SE_FUSED and ps.flat.is_cuda:
                # both FC layers' data gradients in one launch
                call("mmd_se_fc_bwd_fused", pool5[0], rec["gate"], rec["hpre"], ps.w(f"{q}._se_reduce.conv.weight"),
                     ps.w(f"{q}._se_expand.conv.weight"), dpe, dpr, dpooled, 1.0 / HW1, f1.B, blk.cmid, blk.se, pool5, sums1)
            else:
                dh = self._zalloc((f1.B, blk.se))
                call("mmd_se_fc_bwd", pool5[0], rec["gate"], rec["hpre"], rec["pooled"], ps.w(f"{q}._se_reduce.conv.weight"),
                     ps.w(f"{q}._se_expand.conv.weight"), dpe, dpr, dh, dpooled, 1.0 / HW1, None, None, None, None,
                     f1.B, blk.cmid, blk.se, pool5, sums1)
            if SE_WG_BATCH and ps.flat.is_cuda:      # the FC weight gradients of all blocks of the segment in one launch at its end
                self._se_wg.append((dpe, dpr, rec["hpre"], rec["pooled"], *segrads, blk.cmid, blk.se))
            else:
                self._leaf(lambda dpe=dpe, dpr=dpr, hp=rec["hpre"], po=rec["pooled"], gs=segrads, nb=f1.B, cm=blk.cmid, se=blk.se:
                           call("mmd_se_fc_wgrad", dpe, dpr, hp, po, *gs, nb, cm, se))
            f0: Feat = rec.get("f0", inp)
            # BatchNorm-1 backward evaluated in the prologue of the depthwise input-gradient launch (stride-1 blocks with an expand conv and
            # >= 64 channels: 17 of D2's 23): dz1 is never written; the other blocks keep the apply pass
            bn1_in_dw = (BN1_IN_DW and ps.flat.is_cuda and blk.expand != 1 and blk.stride == 1 and blk.cmid >= 64 and self.DW_WG)
            if not bn1_in_dw:
                dz1 = self._bn_bwd(g1, f1.z, a1, f"{q}._bn1", SWISH, M1, blk.cmid, rpi=HW1, mul_bc=rec["gate"],
                                   add_bc=dpooled, sums=sums1)
            if bn1_in_dw:
                b1, a0 = ps.bn(f"{q}._bn1"), rec["bn0"]
                wkey = f"{q}._depthwise_conv.conv.weight"
                g0 = self._alloc(f0.M, f0.C)
                sums0 = self._zalloc((2 * f0.C,), torch.float64)
                call("mmd_dwconv_bwd_data_bn1", g1, f1.z, ps.w(wkey), g0, f0.B, f0.H, f0.W, f0.C, blk.kernel, a1[0], a1[1], a1[2], a1[3],
                     sums1, M1, rec["gate"], dpooled, b1["dgamma"], b1["dbeta"], f0.z, a0[0], a0[1], a0[2], a0[3], sums0,
                     *self._stats_ws(sums0, f0.M, f0.C), ps.g(wkey))
                dz0 = self._bn_bwd(g0, f0.z, rec["bn0"], f"{q}._bn0", SWISH, f0.M, blk.cmid, sums=sums0, lazy=True)
                if dy_read is not None:
                    torch.cuda.current_stream().wait_event(dy_read)
                self._pw_bwd(dz0, inp, f"{q}._expand_conv.conv.weight", blk.cmid, None, True, into=inp)
            elif blk.expand != 1:
                g0, sums0 = self._dw_bwd(dz1, f0, f"{q}._depthwise_conv.conv.weight", blk.kernel, blk.stride, bn_aff=rec["bn0"])
                dz0 = self._bn_bwd(g0, f0.z, rec["bn0"], f"{q}._bn0", SWISH, f0.M, blk.cmid, sums=sums0, lazy=True)
                if dy_read is not None:
                    torch.cuda.current_stream().wait_event(dy_read)
                # the input gradient lands in (on top of) the block input's slot - which may be dy itself, adopted by the skip branch
                # above - and, as that slot's last contribution, carries the previous block's BatchNorm-2 backward sums with it
                self._pw_bwd(dz0, inp, f"{q}._expand_conv.conv.weight", blk.cmid, None, True, into=inp)
            elif blk.idx == 0 and not blk.skip and blk.stride == 1 and slot(inp).t is None:
                # block 0 consumes the stem activation directly and is its only consumer: the stem BN's backward sums ride
                # in this input-gradient launch
                g0, stem_sums = self._dw_bwd(dz1, f0, f"{q}._depthwise_conv.conv.weight", blk.kernel, blk.stride,
                                             bn_aff=(inp.scale, inp.shift, tape["stem"][2], tape["stem"][3]))
                self._acc(inp, g0)
            else:
                g0 = self._dw_bwd(dz1, f0, f"{q}._depthwise_conv.conv.weight", blk.kernel, blk.stride)
                if dy_read is not None:
                    torch.cuda.current_stream().wait_event(dy_read)
                self._acc(inp, g0)
        self._bw["stem_sums"] = stem_sums
        if self._se_wg:
            recs, self._se_wg = self._se_wg, []
            key = tuple((r[0].data_ptr(), r[1].data_ptr(), r[2].data_ptr(), r[3].data_ptr(), r[4].data_ptr()) for r in recs)
            if key not in self._se_wg_tabs:
                if self.arena.frozen:
                    raise RuntimeError("squeeze-excite weight-gradient operands moved after graph capture")
                rows = [[t.data_ptr() for t in r[:8]] + [r[8], r[9]] for r in recs]
                self._se_wg_tabs[key] = torch.tensor(rows, dtype=torch.int64, device=self.device)
            tab, nb = self._se_wg_tabs[key], recs[0][0].shape[0]
            self._leaf(lambda tab=tab, n=len(recs), mx=max(r[8] * r[9] for r in recs), nb=nb: call("mmd_se_fc_wgrad_batched", tab, n, mx, nb))

    def _backward_stem(self):
        ps, tape = self.ps, self.tape
        P = "backbone_net.model"
        stem_sums = self._bw["stem_sums"]
        ximg, stem, mu, istd = tape["stem"]
        s = self._slot(stem)
        Bi, Cin, Hi, Wi = ximg.shape
        dll = _lib.LIB.load() if ps.flat.is_cuda else None
        # (every precision mode: the bf16 rule rounds the operands of the 1x1 convs only - the stem is a 3x3 conv and stays fp32)
        direct = bool(STEM_WG_DIRECT and dll is not None
                      and dll.mmd_stem_conv_bwd_weight_supported(Cin, Hi, Wi, ps.stem_kp, stem.C) == 1)
        if direct:
            # the last grouped flush (the thin 256^2 layers of blocks 0 - 2) starts NOW, beside the stem's BatchNorm backward, and the stem's
            # own weight gradient runs beside it on the second side stream: the step's exposed tail is the longer of the two, not their sum
            self._wg_flush()
        if direct:
            # no im2col matrix (168 MB at 512^2 x 8 channels): the patches are gathered from the image rows while the dz tile is multiplied -
            # and no dz tensor either: the stem BatchNorm's backward is evaluated while that tile is staged (mmd_stem_conv_bwd_weight_bn)
            if stem_sums is None:
                stem_sums = self._zalloc((2 * stem.C,), torch.float64)
                call("mmd_bn_bwd_reduce", s.t, stem.z, stem.scale, stem.shift, mu, istd, SWISH, None, None, None, 0, None, stem_sums, stem.M, stem.C,
                     *self._stats_ws(stem_sums, stem.M, stem.C))
            b0 = ps.bn(f"{P}._bn0")
            ws = self._alloc(int(dll.mmd_stem_wgrad_ws_floats(stem.C)))
            if STEM_WG_SIDE:
                main_stream = torch.cuda.current_stream()
                if self._side is None:
                    self._side = torch.cuda.Stream()
                self._side.wait_event(main_stream.record_event())
                with torch.cuda.stream(self._side):
                    call("mmd_stem_conv_bwd_weight_bn", ximg, s.t, stem.z, ps.g(f"{P}._conv_stem.conv.weight"), ws, Bi, Cin, Hi, Wi, ps.stem_kp, stem.C,
                         stem.scale, stem.shift, mu, istd, stem_sums, stem.M, SWISH, b0["dgamma"], b0["dbeta"])
                main_stream.wait_stream(self._side)
            else:
                # on the MAIN stream, whose chain ends here: in the captured graph the side-stream branch came out serialised BEHIND the grouped
                # flush (rocprofv3 trace: flush 278 us -> fold 46 -> stem 78 -> fold 12, all exposed), the main branch runs beside it
                call("mmd_stem_conv_bwd_weight_bn", ximg, s.t, stem.z, ps.g(f"{P}._conv_stem.conv.weight"), ws, Bi, Cin, Hi, Wi, ps.stem_kp, stem.C,
                     stem.scale, stem.shift, mu, istd, stem_sums, stem.M, SWISH, b0["dgamma"], b0["dbeta"])
        else:
            dz = self._bn_bwd(s.t, stem.z, (stem.scale, stem.shift, mu, istd), f"{P}._bn0", SWISH, stem.M, stem.C, sums=stem_sums)
            with self._wgrad_stream():          # im2col of the input image + weight-gradient GEMM, both off the critical path
                col = self._alloc(stem.M, ps.stem_kp)
                call("mmd_stem_im2col", ximg, col, Bi, Cin, Hi, Wi, ps.stem_kp)
            self._pw_wgrad(dz, col, ps.g(f"{P}._conv_stem.conv.weight"), stem.M, ps.stem_kp, stem.C)
            self._wg_flush()
        if self._wg is not None:
            torch.cuda.current_stream().wait_stream(self._wg)
