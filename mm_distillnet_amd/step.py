"""One distillation step of MM-DistillNet on the HIP engine, capturable as a single hipGraph.

Mirrors the reference's per-step path:
  ModelWithNMSLoss(.Augmented).forward  src/optimization/train_methods.py:310-422 / 436-517
  (ModelWithNMSKDListLoss for kd_mode="list": :84-162)
  loss mixing + backward + step         src/optimization/traditional.py:171-190
with every per-image host loop and `.cpu()` sync replaced by device kernels: teachers' pseudo-labels
(decode, class filter, NMS), the cross-teacher merge + NMS, the focal/smooth-L1 and MTA losses with
their gradients, the student backward, gradient all-reduce (RCCL, student gradients only) and Adam.
"""
from __future__ import annotations

import ctypes
import os

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from . import _lib
from .arch import NetSpec, make_spec
from .engine import Net, Feat, NONE, SWISH, pack_nets
from .store import Arena

call = _lib.call
TEACHER_ORDER = ("rgb", "depth", "thermal")     # ModuleDict insertion order in train.py:123-135
OPT_MODES = {"Adam": 0, "AdamW": 1, "SGD": 2}


@dataclass
class StepConfig:
    image_size: int = 512
    conf_threshold: float = 0.3
    nms_threshold: float = 0.5
    merge_iou: float = 0.5
    T: float = 9.0
    p: float = 2.0
    w_main: float = 1.0
    w_kd: float = 0.005
    lr: float = 1e-4
    b1: float = 0.9
    b2: float = 0.999
    eps: float = 1e-8
    grad_clip: float = -1.0
    optimizer: str = "Adam"            # cfg `optimizer`: SGD | Adam | AdamW (src/optimization/train_methods.py:808-836)
    momentum: float = 0.0              # SGD
    weight_decay: float = 0.0          # SGD: cfg weight_decay; AdamW: torch's default 1e-2 (the reference passes none)
    kd_mode: str = "pairwise"          # "pairwise" = ModelWithNMSLoss(.Augmented); "list" = ModelWithNMSKDListLoss
    valid_prediction_ids: tuple = (6,)  # VOC id of "car" (src/datasets/BaseDataset.py:141-165)
    label_map: Optional[List[int]] = None
    inclusive_nms: bool = False
    max_boxes: int = 0                 # merged pseudo-labels per image handed to the focal loss; 0 = sized for the worst case
    cand_cap: int = 0                  # rows per image of the candidate / pseudo-label arrays; 0 = every anchor (the reference
                                       # has no cap, src/utils/utils.py:179-205), so nothing can overflow
    augment: bool = False              # cfg audio_augmentation_merge (ModelWithNMSLossAugmented.forward augment=True)
    seed: int = 0                      # cfg `seed` (train.py:120 make_reproducible_run): keys the device-side drop-connect draws
    precision: str = "fp32"            # "bf16": 1x1-conv GEMMs of every net on the bf16 MFMA (fp32 accumulate, fp32 tensors)


class DistillEngine:
    def __init__(self, student_spec: NetSpec, teacher_specs: Dict[str, NetSpec], device, cfg: StepConfig,
                 world_size: int = 1, process_group=None):
        self.cfg = cfg
        self.device = device
        self.world_size = world_size
        self.pg = process_group
        self.student = Net(student_spec, device, trainable=True, precision=cfg.precision)
        # (a common stem slot lines the teachers' parameter layouts up behind their stems: the pack mode below)
        slot = max(s.stem_out * ((s.in_channels * 9 + 3) // 4 * 4) for s in teacher_specs.values())
        self.teachers: Dict[str, Net] = {m: Net(teacher_specs[m], device, trainable=False, precision=cfg.precision, stem_slot=slot)
                                         for m in TEACHER_ORDER if m in teacher_specs}
        # Round 4, teacher pack: the frozen teachers share one body architecture, so each layer of all of them runs as ONE launch over
        # 3 x B images with per-group weights instead of three launches on three streams (fewer, fatter launches beside the student's
        # chain: tools/dev/diag_phases.py proxy - student forward + a frozen net at batch 3B 7.28 ms against 8.41 with three nets on three
        # streams).  MMD_NO_PACK=1: one net per stream, the round-3 schedule
        tl = list(self.teachers.values())
        self.pack = (len(tl) > 1 and not os.environ.get("MMD_NO_PACK") and str(device).startswith("cuda")
                     and pack_nets(tl))
        ps = self.student.ps
        n = ps.n_params
        self.exp_avg = torch.zeros(n, device=device)
        self.exp_avg_sq = torch.zeros(n, device=device)
        self.adam_main = torch.zeros(4, device=device)
        self.adam_head = torch.zeros(4, device=device)
        if cfg.optimizer not in OPT_MODES:
            raise Exception(f"Unsupported optimizer {cfg.optimizer}")
        self.opt_mode = OPT_MODES[cfg.optimizer]
        self.lr = float(cfg.lr)
        # device-side hyper-parameters [lr, beta1, beta2, eps, weight_decay, momentum]: a captured graph sees scheduler updates
        self.hyper = torch.tensor([cfg.lr, cfg.b1, cfg.b2, cfg.eps, cfg.weight_decay, cfg.momentum], device=device)
        self.head_active = torch.zeros(1, dtype=torch.int32, device=device)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=device)
        self.ws = Arena(device, 256 << 20)        # loss / pseudo-label workspaces (bump, reset per step)
        self.head_ranges = self._head_ranges()
        nc = student_spec.num_classes
        lm = cfg.label_map if cfg.label_map is not None else list(range(nc))
        self.label_map = torch.tensor(lm, dtype=torch.int32, device=device)
        self.valid_mask = 0
        for i in cfg.valid_prediction_ids:
            self.valid_mask |= 1 << int(i)
        self.cap = int(cfg.cand_cap)     # 0: set to the anchor count at the first step
        self.graph = None
        self._graphs: Dict[str, tuple] = {}       # step variant ("plain" / "aug") -> (static inputs, g_main, g_tail, g_opt, outputs)
        self.concurrent_teachers = True
        self.side_streams: List = []
        self.fork_stream = True if torch.cuda.is_available() and str(device).startswith("cuda") else None
        self.static: Dict[str, torch.Tensor] = {}
        self.out: Dict[str, torch.Tensor] = {}
        self.n_skip = sum(1 for b in student_spec.blocks if b.skip)
        # data-parallel gradient exchange overlapped with the backward: the backward is issued (and captured) in two
        # segments split before backbone block `ar_split`; the gradients of everything behind it (heads, BiFPN, blocks >=
        # ar_split: ~98 % of the buffer, they sit at the tail of the conv region) are reduced on RCCL's stream while
        # the high-resolution early blocks - about half of the backward's time - still run
        self.ar_split: Optional[int] = self._default_split() if world_size > 1 else None
        env = os.environ.get("MMD_AR_SPLIT")
        if env is not None:
            self.ar_split = None if env in ("", "none", "-1") else int(env)
        self._ar_work = None
        self.force_ar = False        # dev aid: issue the collectives on a one-rank group (bench.py MMD_FORCE_DP)
        # MMD_COMM=rccl: the exchange goes through the C ABI's own RCCL communicator (mmd_comm_*: csrc/comm.hip) on a side HIP stream
        # instead of torch.distributed's ProcessGroupNCCL (same library underneath); default: torch.distributed
        self.comm = None
        self.comm_stream = None
        self.keep = torch.tensor([1.0 - b.drop_rate for b in student_spec.blocks if b.skip], device=device).view(-1, 1)
        # device-side drop-connect draws: Philox key (cfg seed, rank-specific like a per-rank torch seed) and [draw counter, injected flag]
        self.drop_seed = (int(getattr(cfg, "seed", 0)) * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03 * (1 + int(os.environ.get("RANK", "0")))) % (1 << 64)
        self.drop_state = torch.zeros(2, dtype=torch.int64, device=device)
        self._drop_injected = False

    # ------------------------------------------------------------------
    def _head_ranges(self):
        """[begin,end) float ranges of the regressor/classifier parameters in the flat buffer."""
        ps = self.student.ps
        conv = [e for e in ps.entries.values() if e.key.startswith(("regressor", "classifier"))]
        b0 = min(e.off for e in conv)
        e0 = ps.n_conv
        bn = [n for n in ps.bn_names if n.startswith(("regressor", "classifier"))]
        lo = min(ps.bn_off[n] for n in bn)
        hi = max(ps.bn_off[n] + ((ps.bn_c[n] + 3) // 4 * 4) for n in bn)
        # heads' conv weights must be the tail of the conv region and their BN channels contiguous
        assert all(e.off >= b0 for e in conv) and all(
            (not k.startswith(("regressor", "classifier"))) or ps.entries[k].off >= b0 for k in ps.order)
        return (b0, e0, ps.gamma_off + lo, ps.gamma_off + hi, ps.beta_off + lo, ps.beta_off + hi)

    def _block_off(self, k: int) -> int:
        ps = self.student.ps
        q = f"backbone_net.model._blocks.{k}."
        return min(e.off for key, e in ps.entries.items() if key.startswith(q))

    def _default_split(self) -> int:
        """First backbone block with more than 2 % of the conv parameters in front of it (D2: block 8, the 64x64 -> 32x32
        stage; the blocks before it own 1.6 % of the gradients and roughly half of the backward's kernel time)."""
        ps = self.student.ps
        k = 1
        for b in self.student.spec.blocks[1:]:
            if self._block_off(b.idx) > 0.02 * ps.n_conv:
                break
            k = b.idx
        return k

    def grad_buckets(self):
        """[(begin, end)] float ranges of the flat gradient buffer per all-reduce phase.  Phase 0 is complete when the
        first backward segment ends, phase 1 (early backbone blocks + all BatchNorm gammas/betas) when the backward ends."""
        ps = self.student.ps
        if self.ar_split is None:
            return [[], [(0, ps.n_params)]]
        o = self._block_off(self.ar_split)
        assert o % 4 == 0
        return [[(o, ps.n_conv)], [(0, o), (ps.n_conv, ps.n_params)]]

    def load(self, student_state, teacher_states: Dict[str, dict]):
        self.student.load_state(student_state)
        for m, net in self.teachers.items():
            net.load_state(teacher_states[m])

    def set_lr(self, lr: float):
        self.lr = float(lr)            # host copy in double (checkpoints carry it; the device copy is its fp32 rounding)
        self.hyper[0] = lr

    def make_drop_scale(self, batch: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """mask/keep per skip block and sample: floor(keep + U[0,1)) / keep (src/YetAnotherEfficientNet.py:173-182).  Host-driven form
        (torch's generator): tests and callers that inject their own masks; the training loop draws on the device, draw_drop_scale."""
        u = torch.rand(self.n_skip, batch, device=self.device, generator=generator)
        return torch.floor(self.keep + u) / self.keep

    def draw_drop_scale(self, out: torch.Tensor):
        """The same draw as ONE HIP launch (csrc/optim.hip drop_scale_kernel: Philox4x32-10, device-side draw counter): captured at the
        head of g_main, so a replayed step issues no ATen launches.  A caller-injected mask (set_drop_scale) is left alone."""
        call("mmd_drop_scale", out, self.keep, self.n_skip, out.shape[1], self.drop_seed, self.drop_state)

    def set_drop_scale(self, drop_scale: Optional[torch.Tensor]):
        """inject (tensor) or release (None) the masks of the captured step: an injected mask survives the graph's own draw."""
        inject = drop_scale is not None
        if inject:
            self.static["drop_scale"].copy_(drop_scale)
        if inject != self._drop_injected:
            self.drop_state[1] = 1 if inject else 0
            self._drop_injected = inject

    # ------------------------------------------------------------------
    def _caps(self, A: int):
        if self.cap <= 0:
            self.cap = A
        return self.cap

    def _nms_ws(self, B: int, nmax: int):
        n = int(_lib.LIB.load().mmd_nms_ws_floats(nmax))
        return self.ws.alloc((B * n,)) if n else None

    def _merge(self, rows_t, cnt_t, B: int, augment: bool):
        """cross-teacher concat + NMS -> (boxes [B,G,5], nbox [B], G).  Up to 4 sources: the teachers in ModuleDict order, then the
        "augmentation" pass of ModelWithNMSKDListLossAugmented."""
        nt, cap = len(rows_t), self.cap
        nmax = nt * cap * (2 if (augment and B >= 2) else 1)
        G = self.cfg.max_boxes if self.cfg.max_boxes > 0 else nmax
        boxes = self.ws.alloc((B, G, 5)); nbox = self.ws.alloc((B,), torch.int32)
        vp = ctypes.c_void_p
        srcs = (vp * nt)(*[t.data_ptr() for t in rows_t]); cnts = (vp * nt)(*[t.data_ptr() for t in cnt_t])
        call("mmd_nms_merge_n", srcs, cnts, nt, float(self.cfg.merge_iou), 1 if self.cfg.inclusive_nms else 0, B, boxes, nbox, G,
             self.mask_ws, self.overflow, 1 if augment else 0, cap, self._nms_ws(B, nmax))
        return boxes, nbox, G

    def _pseudo_labels(self, net: Net, cls, reg, B: int, A: int, S: int):
        ws, cap = self.ws, self._caps(A)
        nc = net.spec.num_classes
        score = ws.alloc((B * A,)); clsid = ws.alloc((B * A,), torch.uint8); flags = ws.alloc((B * A,), torch.uint8)
        over = ws.alloc((B, cap)); cand = ws.alloc((B, cap, 6))
        n_over = ws.alloc((B,), torch.int32); n_keep = ws.alloc((B,), torch.int32)
        call("mmd_decode_filter", cls, reg, net.anchors(S), B, A, nc, float(self.cfg.conf_threshold), self.valid_mask, float(S),
             score, clsid, flags, over, cand, n_over, n_keep, self.overflow, cap)
        rows = ws.alloc((B, cap, 6)); cnt = ws.alloc((B,), torch.int32)
        mask_ws = ws.alloc((B * 1024 * 16,), torch.int64)      # per teacher: teachers run concurrently
        call("mmd_nms_teacher", cand, n_keep, over, self.label_map, float(self.cfg.nms_threshold),
             1 if self.cfg.inclusive_nms else 0, float(S), B, rows, cnt, mask_ws, self.overflow, cap, self._nms_ws(B, cap))
        return rows, cnt

    def _attention(self, net: Net):
        """Spatial attention maps a[b, j] = mean_c f^p of the five BiFPN outputs of `net` in ONE launch: the final cell writes
        them into one pyramid row buffer (engine._bifpn), so the maps are row slices of one vector.  -> (all, [per level])"""
        pyr, fcat = net._pyr, net._fcat
        a = self.ws.alloc((pyr["total"],))
        call("mmd_mta_attention", fcat, a, pyr["total"], fcat.shape[1], float(self.cfg.p))
        return a, [a[pyr["row0"][l]:pyr["row0"][l] + pyr["rows"][l]] for l in range(len(pyr["rows"]))]

    def labels_from_rows(self, per_teacher: List[List], A: int) -> List[tuple]:
        """Host pseudo-labels -> the device form `step_body(teacher_labels=...)` takes: per teacher a list of B arrays [n,6]
        (x1, y1, x2, y2, score, class: what `logits_to_ground_truth(include_scores=True)` returns,
        src/utils/utils.py:234-324) -> (rows [B, cap, 6] fp32, count [B] int32)."""
        import numpy as np
        self._caps(A)
        out = []
        for lab in per_teacher:
            B = len(lab)
            rows = torch.zeros(B, self.cap, 6)
            cnt = torch.zeros(B, dtype=torch.int32)
            for i, a in enumerate(lab):
                a = np.asarray(a, dtype=np.float32).reshape(-1, 6)
                if a.shape[0] > self.cap:
                    raise ValueError("more pseudo-labels than the candidate arena holds")
                rows[i, :a.shape[0]] = torch.from_numpy(a)
                cnt[i] = a.shape[0]
            out.append((rows.to(self.device), cnt.to(self.device)))
        return out

    def step_body(self, batch: Dict[str, torch.Tensor], drop_scale: Optional[torch.Tensor], teacher_labels: Optional[List[tuple]] = None,
                  train: bool = True):
        """Issues the whole step on the current stream.  batch tensors are NCHW fp32 on device.
        teacher_labels (optional): per teacher (rows [B, cap, 6], count [B]) pseudo-labels computed elsewhere (cached labels of the
        frozen teachers, or a reference run's labels in the parity tests); the teachers' own decode + NMS is then skipped, their
        forward still feeds the MTA loss.
        train=False: the losses of `validate()` (src/optimization/train_methods.py:1083-1185): eval-mode student (running
        BatchNorm statistics, no drop-connect), same pseudo-labels and loss terms, no backward."""
        cfg = self.cfg
        st = self.student
        S = cfg.image_size
        B = batch["audio"].shape[0]
        # validate() upstream calls model(..., validate=True) with `augment` left at its default False
        # (src/optimization/train_methods.py:1083-1185): the validation losses never see the audio merge / feature averaging / label merge
        aug = bool(cfg.augment and train)
        self.ws.reset()
        self.mask_ws = self.ws.alloc((B * 1024 * 16,), torch.int64)
        if self.fork_stream is None:
            self.concurrent_teachers = False
        fork_event = None
        if self.concurrent_teachers:
            # teachers fork from here, before the student forward is enqueued
            fork_event = torch.cuda.current_stream().record_event()
        st.begin_step()
        audio = batch["audio"]
        if aug and B >= 2:      # merge_batch_0_1: image 1 <- log10(a0^10 + a1^10), out of place
            merged = st._alloc(*audio.shape)
            call("mmd_audio_merge01", audio, merged, audio[0].numel(), B)
            audio = merged
        st.mark_block, st.mark_event = int(os.environ.get("MMD_PACK_STAGGER", "-1")), None      # (dev: hold the teacher pack until the student passed block k)
        cls_s, reg_s, feats_s = st.forward(audio, train=train, drop_scale=drop_scale if train else None)
        A = cls_s.shape[1]
        self._caps(A)
        nlv = len(feats_s)
        _, a_s = self._attention(st)
        # gradient w.r.t. the student's maps, one vector over the pyramid rows (zeroed: teachers accumulate into it
        # with atomics, and the padding rows of a pyramid must read as zero in the backward)
        da_all = self.ws.alloc((st._pyr["total"],))
        call("mmd_memset_async", da_all, 0, da_all.numel() * 4)
        da = [da_all[st._pyr["row0"][l]:st._pyr["row0"][l] + st._pyr["rows"][l]] for l in range(nlv)]
        nt = len(self.teachers) + (1 if (cfg.kd_mode == "list" and batch.get("aug_rgb") is not None) else 0)
        kd = self.ws.alloc((nt if cfg.kd_mode == "pairwise" else 1, nlv))
        call("mmd_memset_async", kd, 0, kd.numel() * 4)
        # the three frozen teachers are independent of each other and of the student forward: issue them on
        # side streams (parallel branches of the captured graph) so their many small kernels fill the chip together
        rows_t, cnt_t, att_t = [], [], []
        main_stream = torch.cuda.current_stream()
        concurrent = self.concurrent_teachers and torch.cuda.is_available()
        if concurrent and not self.side_streams:
            self.side_streams = [torch.cuda.Stream() for _ in self.teachers]
        passes = [(mod, net, ti, None) for ti, (mod, net) in enumerate(self.teachers.items())]
        if cfg.kd_mode == "list" and batch.get("aug_rgb") is not None:
            # ModelWithNMSKDListLossAugmented.forward(augment=True) (src/optimization/train_methods.py:73-110): the RGB teacher once more,
            # on `label` = RGB frames of the recordings whose audio was mixed into this batch; its pseudo-labels are concatenated after
            # the teachers' and its features join the MTA list.  Same net object, same side stream: the second pass starts after the
            # first one's labels / attention maps have been taken (stream order), so it may reuse the net's arena.
            if "rgb" not in self.teachers:
                raise ValueError("the augmentation pass runs the RGB teacher")
            passes.append(("augmentation", self.teachers["rgb"], list(self.teachers).index("rgb"), batch["aug_rgb"]))
        # Staggered teachers: the frozen nets have the same layer sequence, so three streams started together run in lockstep - three
        # MFMA-bound GEMMs at once, then three HBM-bound depthwise convs at once.  Teacher i+1 therefore starts once teacher i is past
        # the high-resolution stages (its second stride-2 block: D2 block 5): 20.3-20.7 -> 19.9-20.1 ms/step in alternating runs
        # (profiles/r02_notes.md).  MMD_STAGGER=<block> overrides, -1 disables.  Holding teacher 0 behind the student as well measured
        # slower (20.3-20.5).
        s2 = [b.idx for b in next(iter(self.teachers.values())).spec.blocks if b.stride == 2]
        stagger = int(os.environ.get("MMD_STAGGER", str(s2[1] if len(s2) > 1 else -1)))
        prev_net = None
        # teacher pack (round 4): every layer of all the frozen teachers as ONE launch over G x B images with per-group weights, on one side
        # stream.  Needs whole 128-row tiles per group on every pyramid level (the smallest, S/128 squared, decides) and no per-teacher
        # feature surgery (the augmented variant averages images 0 / 1 of each teacher's maps)
        G = len(self.teachers)
        # (round 5: the smallest level needs whole 32-row tiles per net only - the heads run a level that misses the 128-row tiles as plain
        # launches on the skinny kernel, engine.Net._head - so D4 / 768^2 at B = 8, 288 rows per net on its 6 x 6 level, packs too)
        # (the 32-row relaxation holds only where EVERY per-net launch on that level is covered - the heads split it off themselves,
        # the BiFPN nodes only on the whole-node kernel, which has no row-tile constraint; a width without one (88 / 288 / 384: D1, D5 - D7), the
        # wide form past its row cap or MMD_NO_NODE_FUSE send that level's nodes through mmd_pwconv_fwd, whose 64- / 128-row kernels refuse
        # a tile that straddles two nets: those geometries keep the whole-128-row-tile rule, i.e. run one net per stream as before)
        rows_small = B * (S // 128) ** 2
        net_f = next(iter(self.teachers.values()))
        node_whole = bool(net_f.FUSE_NODE and net_f.ps.flat.is_cuda and
                          _lib.LIB.load().mmd_bifpn_node_fused_supported(net_f.spec.fpn_w) == 1 and
                          (net_f.spec.fpn_w <= 160 or G * rows_small <= net_f.NODE_FUSE_WIDE_MAXROWS))
        use_pack = bool(self.pack and G > 1 and not aug and S % 128 == 0 and
                        (rows_small % 128 == 0 or (rows_small % 32 == 0 and node_whole)))
        npk = min(G, max(2, int(os.environ.get("MMD_PACK_SPLIT", G))))      # (dev: pack the first npk teachers, the others on their own streams)
        if use_pack:
            nets = list(self.teachers.values())[:npk]
            net0 = nets[0]
            side = self.side_streams[0] if concurrent else main_stream
            if concurrent:
                side.wait_event(fork_event)
                if st.mark_event is not None:
                    side.wait_event(st.mark_event)
            with torch.cuda.stream(side):
                net0.begin_step()
                cls_p, reg_p, feats_p = net0.forward([batch[m] for m in list(self.teachers)[:npk]], train=False, pack=nets)
                if teacher_labels is None:
                    rows_p, cnt_p = self._pseudo_labels(net0, cls_p, reg_p, npk * B, A, S)
                _, a_lv = self._attention(net0)
            for gi in range(npk):
                if teacher_labels is not None:
                    r, c = teacher_labels[gi]
                else:
                    r, c = rows_p[gi * B:(gi + 1) * B], cnt_p[gi * B:(gi + 1) * B]
                rows_t.append(r); cnt_t.append(c)
                att_t.append([a_lv[l][gi * B * f.H * f.W:(gi + 1) * B * f.H * f.W] for l, f in enumerate(feats_p)])
            passes = [p_ for p_ in passes if p_[3] is not None or p_[2] >= npk]      # (the KD-list variant's extra RGB pass still runs on its own)
        for pi, (mod, net, si, xin) in enumerate(passes):
            ti = si if xin is None else G
            side = self.side_streams[si] if concurrent else main_stream
            if concurrent and xin is None:
                side.wait_event(fork_event)
                if stagger >= 0 and prev_net is not None and prev_net.mark_event is not None:
                    side.wait_event(prev_net.mark_event)
                net.mark_block, net.mark_event = stagger, None
                prev_net = net
            with torch.cuda.stream(side):
                net.begin_step()
                cls_t, reg_t, feats_t = net.forward(xin if xin is not None else (audio if mod == "audio" else batch[mod]), train=False)
                if aug and B >= 2:      # average_batch_0_1 on the (already consumed by the heads) feature maps
                    for f in feats_t:
                        call("mmd_avg_image01", f.z, f.H * f.W * f.C)
                if teacher_labels is not None:
                    r, c = teacher_labels[ti]
                else:
                    r, c = self._pseudo_labels(net, cls_t, reg_t, B, A, S)
                _, at = self._attention(net)
            rows_t.append(r); cnt_t.append(c); att_t.append(at)
        if concurrent:
            for side in self.side_streams:
                main_stream.wait_stream(side)
        # every (level, teacher) KL of the step in one launch (pairwise: nt x 5 losses; list: 5), then one launch for d attention / d f
        vp = ctypes.c_void_p
        p_as = (vp * nlv)(*[t.data_ptr() for t in a_s])
        p_at = (vp * (nt * nlv))(*[att_t[ti][l].data_ptr() for ti in range(nt) for l in range(nlv)])
        p_da = (vp * nlv)(*[t.data_ptr() for t in da])
        hw = (ctypes.c_int * nlv)(*[f.H * f.W for f in feats_s])
        call("mmd_mta_kl_multi", p_as, p_at, p_da, hw, nlv, nt, 0 if cfg.kd_mode == "pairwise" else 1, B, float(cfg.T), kd,
             float(cfg.w_kd))
        if train:
            d_all = st._alloc(st._pyr["total"], feats_s[0].C)
            call("mmd_mta_attention_bwd", st._fcat, da_all, d_all, st._pyr["total"], feats_s[0].C, float(cfg.p), 0)
            dfe = [d_all[st._pyr["row0"][l]:st._pyr["row0"][l] + st._pyr["rows"][l]] for l in range(nlv)]
        # cross-teacher merge -> annotations
        boxes, nbox, G = self._merge(rows_t, cnt_t, B, aug)
        # focal + smooth-L1 with gradients w.r.t. (pre-sigmoid) classifier logits and regression
        nc = st.spec.num_classes
        assign = self.ws.alloc((B * A,), torch.int32); npos = self.ws.alloc((B,), torch.int32)
        acc = self.ws.alloc((2 * B,), torch.float64); main = self.ws.alloc((2,))
        dcls = st._alloc(B, A, nc); dreg = st._alloc(B, A, 4)
        call("mmd_focal_loss", cls_s, reg_s, st.anchors(S), boxes, nbox, G, B, A, nc, assign, npos, acc, main, dcls, dreg,
             float(cfg.w_main), 1, self.head_active if train else self.ws.alloc((1,), torch.int32))
        # backward + optimizer
        if train and not _lib.dev_switch("MMD_DEV_NO_BWD"):      # (MMD_DEV_NO_BWD=1: timing experiment - forward + losses only)
            call("mmd_memset_async", st.ps.grad, 0, st.ps.grad.numel() * 4)
            st.backward(dcls, dreg, dfe, stop_before=self.ar_split, dfeat_pyr=d_all)
        self.out = {"reg": main[0:1], "cls": main[1:2], "kd": kd, "boxes": boxes, "nbox": nbox,
                    "cls_s": cls_s, "reg_s": reg_s, "feats_s": feats_s, "rows_t": rows_t, "cnt_t": cnt_t}
        return self.out

    def backward_tail(self):
        """Second backward segment (backbone blocks < ar_split + stem); a no-op when the backward is not split."""
        if self.ar_split is not None:
            self.student.backward_finish(self.ar_split)

    def init_comm(self, rank: int):
        """Create the C-ABI RCCL communicator (MMD_COMM=rccl).  The 128-byte rendezvous token is made on rank 0 and handed out through
        the already-initialised torch.distributed group (any backend: it only carries these 128 bytes)."""
        import torch.distributed as dist
        dll = _lib.LIB.load()
        tok = (ctypes.c_char * 128)()
        if rank == 0:
            rc = dll.mmd_comm_unique_id(ctypes.cast(tok, ctypes.c_void_p))
            if rc != 0:
                raise RuntimeError(f"mmd_comm_unique_id failed with status {rc}")
        box = [bytes(tok)]
        if self.world_size > 1:
            dist.broadcast_object_list(box, src=0, group=self.pg)
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(box[0], 128)
        rc = dll.mmd_comm_init(ctypes.cast(ctypes.pointer(h), ctypes.c_void_p), rank, self.world_size, ctypes.cast(buf, ctypes.c_void_p))
        if rc != 0:
            raise RuntimeError(f"mmd_comm_init failed with status {rc}")
        self.comm = h
        self.comm_stream = torch.cuda.Stream()

    def close_comm(self):
        if self.comm is not None:
            torch.cuda.synchronize()
            _lib.LIB.load().mmd_comm_destroy(self.comm)
            self.comm = None

    def _comm_allreduce(self, phase):
        """C-ABI path: buckets of `phase` on the communicator's stream, ordered after everything enqueued on the compute stream so
        far; phase 1 also makes the compute stream wait for the whole exchange."""
        dll = _lib.LIB.load()
        g = self.student.ps.grad
        buckets = self.grad_buckets()
        cur = torch.cuda.current_stream()
        self.comm_stream.wait_event(cur.record_event())
        sp = self.comm_stream.cuda_stream

        def red(t, n, dtype, op):
            rc = dll.mmd_comm_allreduce_bucket(self.comm, ctypes.c_void_p(t.data_ptr()), n, dtype, op, ctypes.c_void_p(sp))
            if rc != 0:
                raise RuntimeError(f"mmd_comm_allreduce_bucket failed with status {rc}")
        if phase in (0, None):
            for b, e in buckets[0]:
                red(g[b:e], e - b, 0, 0)
            red(self.head_active, 1, 1, 1)
        if phase in (1, None):
            for b, e in buckets[1]:
                red(g[b:e], e - b, 0, 0)
            cur.wait_event(self.comm_stream.record_event())

    def allreduce_grads(self, phase: Optional[int] = None):
        """RCCL all-reduce(sum) of the flat student gradient buffer (teachers are frozen: nothing else is
        exchanged).  The 1/world average is folded into the optimizer's grad_scale.  head_active is reduced
        with max so that every rank gates the same parameter ranges (ranks may disagree on whether their
        batch had pseudo-labels).

        phase 0: issued after the first backward segment, asynchronously - the collective runs on the process group's
        own stream (ordered after everything enqueued so far) next to the second backward segment.  phase 1: the
        remaining ranges after the backward's end; it also makes the current stream wait for phase 0.  phase None: both."""
        if self.world_size <= 1 and not self.force_ar:
            return
        if self.comm is not None:
            return self._comm_allreduce(phase)
        import torch.distributed as dist
        g = self.student.ps.grad
        buckets = self.grad_buckets()
        if phase in (0, None):
            self._ar_work = [dist.all_reduce(g[b:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                             for b, e in buckets[0]]
            self._ar_work.append(dist.all_reduce(self.head_active, op=dist.ReduceOp.MAX, group=self.pg, async_op=True))
        if phase in (1, None):
            work = (self._ar_work or []) + [dist.all_reduce(g[b:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                                            for b, e in buckets[1]]
            for w in work:
                w.wait()
            self._ar_work = None

    def optimizer_body(self):
        cfg, ps = self.cfg, self.student.ps
        gs = 1.0 / self.world_size
        if cfg.grad_clip > 0:
            if self.world_size > 1:
                ps.grad.mul_(gs)
                gs = 1.0
            ws = self.ws.alloc((1,), torch.float64)
            call("mmd_clip_grad_norm", ps.grad, ps.grad.numel(), float(cfg.grad_clip), ws)
        r = self.head_ranges
        if self.opt_mode == 0:
            call("mmd_adam_step_gated", ps.flat, ps.grad, self.exp_avg, self.exp_avg_sq, self.adam_main, self.adam_head,
                 self.hyper, self.head_active, r[0], r[1], r[2], r[3], r[4], r[5], float(gs), ps.n_params)
        else:
            call("mmd_opt_step_gated", self.opt_mode, ps.flat, ps.grad, self.exp_avg, self.exp_avg_sq, self.adam_main, self.adam_head,
                 self.hyper, self.head_active, r[0], r[1], r[2], r[3], r[4], r[5], float(gs), ps.n_params)
        self.student.refresh_wt()

    # ------------------------------------------------------------------ eager / graph drivers
    def step(self, batch: Dict[str, torch.Tensor], drop_scale: Optional[torch.Tensor] = None):
        """Eager step (no graph): forward, losses, backward, all-reduce, Adam."""
        B = batch["audio"].shape[0]
        if drop_scale is None:
            if self._drop_injected:
                # an earlier replay(..., drop_scale=mask) / set_drop_scale(mask) left the device flag up: mmd_drop_scale would then leave the
                # fresh buffer below unfilled (garbage drop-connect scales, no error) - release the injection before drawing
                self.set_drop_scale(None)
            drop_scale = torch.empty(self.n_skip, B, device=self.device)
            self.draw_drop_scale(drop_scale)
        out = self.step_body(batch, drop_scale)
        self.allreduce_grads(0)
        self.backward_tail()
        self.allreduce_grads(1)
        self.optimizer_body()
        return out

    @staticmethod
    def _variant(batch) -> str:
        return "aug" if (batch is not None and batch.get("aug_rgb") is not None) else "plain"

    def capture(self, batch: Dict[str, torch.Tensor]):
        """Warm up eagerly (sizes the arenas), then capture forward+loss+backward and the optimizer as
        hipGraphs (the backward in two segments when world_size > 1); the gradient all-reduce is launched eagerly between
        them (RCCL is not captured) and overlaps with the second backward segment.
        One set of graphs per step variant: "plain", and "aug" when the batch carries `aug_rgb` (the extra RGB-teacher pass of
        ModelWithNMSKDListLossAugmented); replay() picks by the batch's keys and captures a missing variant on first use."""
        B = batch["audio"].shape[0]
        arenas = [self.ws, self.student.arena, self.student.zarena] + [a for n in self.teachers.values() for a in (n.arena, n.zarena)]
        for arena in arenas:
            arena.frozen = False             # a later variant may need more workspace: chunks are only ever appended
        self.static = {k: v.clone() for k, v in batch.items() if v is not None}
        self.static["drop_scale"] = self.make_drop_scale(B)
        torch.cuda.synchronize()
        # snapshot so the warm-up steps do not change the training trajectory
        ps = self.student.ps
        snap = [t.clone() for t in (ps.flat, ps.rmean, ps.rvar, ps.nbt, self.exp_avg, self.exp_avg_sq, self.adam_main,
                                    self.adam_head, self.head_active)]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self.step_body(self.static, self.static["drop_scale"])
                self.backward_tail()
                self.optimizer_body()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for arena in arenas:
            arena.frozen = True
        # thread_local: with a process group alive, the RCCL watchdog thread polls its own events; under the default
        # "global" mode such a call from another thread would invalidate the capture
        self.g_main = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_main, capture_error_mode="thread_local"):
            self.draw_drop_scale(self.static["drop_scale"])      # (no-op while a mask is injected: set_drop_scale)
            self.step_body(self.static, self.static["drop_scale"])
        out = self.out
        self.g_tail = None
        if self.ar_split is not None:
            self.g_tail = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_tail, capture_error_mode="thread_local"):
                self.backward_tail()
        self.g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_opt, capture_error_mode="thread_local"):
            self.optimizer_body()
        for dst, src in zip((ps.flat, ps.rmean, ps.rvar, ps.nbt, self.exp_avg, self.exp_avg_sq, self.adam_main,
                             self.adam_head, self.head_active), snap):
            dst.copy_(src)
        self.student.refresh()
        torch.cuda.synchronize()
        self.graph = True
        self._graphs[self._variant(batch)] = (self.static, self.g_main, self.g_tail, self.g_opt, out)

    def replay(self, batch: Optional[Dict[str, torch.Tensor]] = None, drop_scale: Optional[torch.Tensor] = None):
        if batch is not None:
            key = self._variant(batch)
            if key not in self._graphs:
                self.capture(batch)
            self.static, self.g_main, self.g_tail, self.g_opt, self.out = self._graphs[key]
            for k, v in batch.items():
                if v is not None:
                    self.static[k].copy_(v, non_blocking=True)
        if drop_scale is not None or self._drop_injected:
            self.set_drop_scale(drop_scale)
        self.g_main.replay()
        self.allreduce_grads(0)
        if self.g_tail is not None:
            self.g_tail.replay()
        self.allreduce_grads(1)
        self.g_opt.replay()
        return self.out

    @torch.no_grad()
    def predict(self, batch: Dict[str, torch.Tensor]):
        """Evaluation path (get_predictions_multiteacher): student detections [n,6] per image from an
        eval-mode forward, and the merged multi-teacher pseudo ground truth [m,5] per image, as numpy lists."""
        cfg, S = self.cfg, self.cfg.image_size
        B = batch["audio"].shape[0]
        self.ws.reset()
        self.mask_ws = self.ws.alloc((B * 1024 * 16,), torch.int64)
        st = self.student
        st.refresh()
        st.begin_step()
        cls_s, reg_s, _ = st.forward(batch["audio"], train=False)
        A = cls_s.shape[1]
        rows_s, cnt_s = self._pseudo_labels(st, cls_s, reg_s, B, A, S)
        rows_t, cnt_t = [], []
        for mod, net in self.teachers.items():
            net.begin_step()
            cls_t, reg_t, _ = net.forward(batch[mod], train=False)
            r, c = self._pseudo_labels(net, cls_t, reg_t, B, A, S)
            rows_t.append(r); cnt_t.append(c)
        boxes, nbox, G = self._merge(rows_t, cnt_t, B, False)
        torch.cuda.synchronize()
        cs, nb = cnt_s.cpu().tolist(), nbox.cpu().tolist()
        preds = [rows_s[i, :cs[i]].cpu().numpy() for i in range(B)]
        labels = [boxes[i, :nb[i]].cpu().numpy() for i in range(B)]
        return preds, labels

    @torch.no_grad()
    def eval_losses(self, batch: Dict[str, torch.Tensor], teacher_labels: Optional[List[tuple]] = None):
        """-> (reg, cls, kd_sum) python floats of one validation batch (eval-mode student; validate() upstream,
        src/optimization/train_methods.py:1135-1150: the sums over the step module's loss lists).  teacher_labels: as step_body."""
        self.student.refresh()
        out = self.step_body(batch, None, teacher_labels=teacher_labels, train=False)
        return out["reg"].item(), out["cls"].item(), out["kd"].sum().item()

    def check_overflow(self):
        if int(self.overflow.item()):
            raise RuntimeError("pseudo-label capacity exceeded (cfg cand_cap = %d rows / max_boxes = %d per image; 0 = unlimited)"
                               % (self.cfg.cand_cap, self.cfg.max_boxes))
