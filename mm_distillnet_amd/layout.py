"""State-dict layout of the detector: ordered (key, shape, kind) rows.

Key names and ordering restate what `YetAnotherEfficientDet(...).state_dict()` yields in the
reference (module registration order bifpn -> regressor -> classifier -> backbone_net,
src/YetAnotherEfficientDet.py:639-655; checkpoint-key compatibility is part of the drop-in boundary,
SURVEY.md §8(b)).  tests/golden/state_keys_d2.json pins it against the reference.

kind in {"pw", "dw", "stem", "se_w", "bias", "bn_w", "bn_b", "bn_rm", "bn_rv", "bn_nbt", "fuse"}.
"""
from __future__ import annotations

from typing import List, Tuple

from .arch import NetSpec

Row = Tuple[str, Tuple[int, ...], str]


def _bn(p: str, c: int) -> List[Row]:
    return [(p + ".weight", (c,), "bn_w"), (p + ".bias", (c,), "bn_b"),
            (p + ".running_mean", (c,), "bn_rm"), (p + ".running_var", (c,), "bn_rv"),
            (p + ".num_batches_tracked", (), "bn_nbt")]


def _sep(p: str, cin: int, cout: int, norm: bool) -> List[Row]:
    rows = [(p + ".depthwise_conv.conv.weight", (cin, 1, 3, 3), "dw"),
            (p + ".pointwise_conv.conv.weight", (cout, cin, 1, 1), "pw"),
            (p + ".pointwise_conv.conv.bias", (cout,), "bias")]
    if norm:
        rows += _bn(p + ".bn", cout)
    return rows


def state_layout(spec: NetSpec) -> List[Row]:
    W = spec.fpn_w
    rows: List[Row] = []
    for c in range(spec.fpn_cells):
        p = f"bifpn.{c}"
        rows += [(f"{p}.p6_w1", (2,), "fuse"), (f"{p}.p5_w1", (2,), "fuse"), (f"{p}.p4_w1", (2,), "fuse"),
                 (f"{p}.p3_w1", (2,), "fuse"), (f"{p}.p4_w2", (3,), "fuse"), (f"{p}.p5_w2", (3,), "fuse"),
                 (f"{p}.p6_w2", (3,), "fuse"), (f"{p}.p7_w2", (2,), "fuse")]
        for n in ["conv6_up", "conv5_up", "conv4_up", "conv3_up", "conv4_down", "conv5_down", "conv6_down",
                  "conv7_down"]:
            rows += _sep(f"{p}.{n}", W, W, True)
        if c == 0:
            c3, c4, c5 = spec.p345
            for n, ci in [("p5_down_channel", c5), ("p4_down_channel", c4), ("p3_down_channel", c3),
                          ("p5_to_p6", c5), ("p4_down_channel_2", c4), ("p5_down_channel_2", c5)]:
                rows += [(f"{p}.{n}.0.conv.weight", (W, ci, 1, 1), "pw"), (f"{p}.{n}.0.conv.bias", (W,), "bias")]
                rows += _bn(f"{p}.{n}.1", W)
    for hname, per_anchor in [("regressor", 4), ("classifier", spec.num_classes)]:
        for i in range(spec.head_layers):
            rows += _sep(f"{hname}.conv_list.{i}", W, W, False)
        for lvl in range(5):
            for i in range(spec.head_layers):
                rows += _bn(f"{hname}.bn_list.{lvl}.{i}", W)
        rows += _sep(f"{hname}.header", W, spec.num_anchors * per_anchor, False)
    p = "backbone_net.model"
    rows += [(f"{p}._conv_stem.conv.weight", (spec.stem_out, spec.in_channels, 3, 3), "stem")]
    rows += _bn(f"{p}._bn0", spec.stem_out)
    for b in spec.blocks:
        q = f"{p}._blocks.{b.idx}"
        if b.expand != 1:
            rows += [(f"{q}._expand_conv.conv.weight", (b.cmid, b.cin, 1, 1), "pw")]
            rows += _bn(f"{q}._bn0", b.cmid)
        rows += [(f"{q}._depthwise_conv.conv.weight", (b.cmid, 1, b.kernel, b.kernel), "dw")]
        rows += _bn(f"{q}._bn1", b.cmid)
        rows += [(f"{q}._se_reduce.conv.weight", (b.se, b.cmid, 1, 1), "se_w"),
                 (f"{q}._se_reduce.conv.bias", (b.se,), "bias"),
                 (f"{q}._se_expand.conv.weight", (b.cmid, b.se, 1, 1), "se_w"),
                 (f"{q}._se_expand.conv.bias", (b.cmid,), "bias")]
        rows += [(f"{q}._project_conv.conv.weight", (b.cout, b.cmid, 1, 1), "pw")]
        rows += _bn(f"{q}._bn2", b.cout)
    return rows


def param_rows(spec: NetSpec) -> List[Row]:
    """Trainable parameters only (what `named_parameters()` yields), same order."""
    return [r for r in state_layout(spec) if r[2] not in ("bn_rm", "bn_rv", "bn_nbt")]
