"""Architecture tables for the EfficientDet family used by MM-DistillNet.

This is the build's own description of the network that the reference constructs in
`src/YetAnotherEfficientNet.py:150-170,317-343,492-604` (EfficientNet scaling rules and block
strings) and `src/YetAnotherEfficientDet.py:605-655` (BiFPN width / repeats / head depth tables).
Only plain data comes out of here; the engine (engine.py) and the nn.Module facade (model.py)
both consume it.  State-dict key names follow the reference so checkpoints interchange.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Tuple

# (width, depth) multipliers of efficientnet-b0..b8  (reference: YetAnotherEfficientNet.py:226-241)
_EFFNET_COEF = {
    0: (1.0, 1.0), 1: (1.0, 1.1), 2: (1.1, 1.2), 3: (1.2, 1.4), 4: (1.4, 1.8),
    5: (1.6, 2.2), 6: (1.8, 2.6), 7: (2.0, 3.1), 8: (2.2, 3.6),
}
# base stages: (repeat, kernel, stride, expand, in, out)   (YetAnotherEfficientNet.py:321-326)
_BASE_STAGES = [
    (1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
    (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320),
]
# EfficientDet compound tables (YetAnotherEfficientDet.py:611-629)
BACKBONE_COEF = [0, 1, 2, 3, 4, 5, 6, 6]
FPN_FILTERS = [64, 88, 112, 160, 224, 288, 384, 384]
FPN_REPEATS = [3, 4, 5, 6, 7, 7, 8, 8]
INPUT_SIZES = [512, 640, 768, 896, 1024, 1280, 1280, 1536]
HEAD_REPEATS = [3, 3, 3, 4, 4, 4, 5, 5]
ANCHOR_SCALE = [4., 4., 4., 4., 4., 4., 4., 5.]
P345_CHANNELS = {0: [40, 112, 320], 1: [40, 112, 320], 2: [48, 120, 352], 3: [48, 136, 384],
                 4: [56, 160, 448], 5: [64, 176, 512], 6: [72, 200, 576], 7: [72, 200, 576]}

BN_EPS = 1e-3
BN_MOMENTUM = 0.01
DROP_CONNECT_RATE = 0.2
SE_RATIO = 0.25
FUSE_EPS = 1e-4
NUM_ANCHORS = 9


def round_filters(filters: float, width: float, divisor: int = 8) -> int:
    filters = filters * width
    new = max(divisor, int(filters + divisor / 2) // divisor * divisor)
    if new < 0.9 * filters:
        new += divisor
    return int(new)


def round_repeats(repeats: int, depth: float) -> int:
    return int(math.ceil(depth * repeats))


@dataclass
class BlockSpec:
    idx: int
    kernel: int
    stride: int
    cin: int
    cout: int
    expand: int
    se: int            # squeezed channels
    skip: bool         # identity skip + drop-connect
    drop_rate: float   # drop-connect probability in training

    @property
    def cmid(self) -> int:
        return self.cin * self.expand


@dataclass
class NetSpec:
    compound_coef: int
    in_channels: int
    num_classes: int
    stem_out: int
    blocks: List[BlockSpec]
    taps: List[int]                 # block indices whose outputs are p3, p4, p5
    fpn_w: int
    fpn_cells: int
    head_layers: int
    anchor_scale: float
    p345: List[int] = field(default_factory=list)

    @property
    def num_anchors(self) -> int:
        return NUM_ANCHORS


def make_spec(compound_coef: int = 2, in_channels: int = 3, num_classes: int = 20) -> NetSpec:
    width, depth = _EFFNET_COEF[BACKBONE_COEF[compound_coef]]
    blocks: List[BlockSpec] = []
    for (r, k, s, e, i, o) in _BASE_STAGES:
        ci, co = round_filters(i, width), round_filters(o, width)
        for j in range(round_repeats(r, depth)):
            cin = ci if j == 0 else co
            stride = s if j == 0 else 1
            # reference quirk: the first block of a stage carries stride as a list, so `stride == 1`
            # is False there and it never takes the skip (YetAnotherEfficientNet.py:272,481,536);
            # cin != cout on those blocks anyway.
            skip = (j > 0) and (cin == co)
            blocks.append(BlockSpec(idx=len(blocks), kernel=k, stride=stride, cin=cin, cout=co, expand=e,
                                    se=max(1, int(cin * SE_RATIO)), skip=skip, drop_rate=0.0))
    n = len(blocks)
    for b in blocks:
        b.drop_rate = DROP_CONNECT_RATE * float(b.idx) / n
    # taps (YetAnotherEfficientDet.py:560-572): output of the block before each stride-2 block plus
    # the last block -> 5 maps at S/2,S/4,S/8,S/16,S/32; `[1:]` and the `_` in
    # YetAnotherEfficientDet.forward (:665) leave the last three as p3,p4,p5.
    taps = _taps(blocks)
    taps = taps[2:]
    p345 = [blocks[t].cout for t in taps]
    assert p345 == P345_CHANNELS[compound_coef], (p345, P345_CHANNELS[compound_coef])
    return NetSpec(compound_coef=compound_coef, in_channels=in_channels, num_classes=num_classes,
                   stem_out=round_filters(32, width), blocks=blocks, taps=taps,
                   fpn_w=FPN_FILTERS[compound_coef], fpn_cells=FPN_REPEATS[compound_coef],
                   head_layers=HEAD_REPEATS[compound_coef], anchor_scale=ANCHOR_SCALE[compound_coef],
                   p345=p345)


def _taps(blocks: List[BlockSpec]) -> List[int]:
    fm = []
    for b in blocks:
        if b.stride == 2:
            fm.append(b.idx - 1)
        elif b.idx == len(blocks) - 1:
            fm.append(b.idx)
    return fm


def same_pad(n: int, k: int, s: int) -> Tuple[int, int, int]:
    """TF-SAME padding (YetAnotherEfficientNet.py:51-62): returns (lo, hi, out_len)."""
    out = -(-n // s)
    extra = (out - 1) * s - n + k
    lo = extra // 2
    return lo, extra - lo, out


def pyramid_sizes(image_size: int) -> List[int]:
    return [image_size // (2 ** l) for l in (3, 4, 5, 6, 7)]


def num_anchor_rows(image_size: int) -> int:
    return sum(s * s for s in pyramid_sizes(image_size)) * NUM_ANCHORS
