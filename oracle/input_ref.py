"""ORACLE (test infrastructure only): the reference's per-sample input transforms on the CPU, numpy float64.

Restates `Normalizer` / `Resizer` (src/datasets/transformations.py:315-330, 407-467), the thermal clamp + min-max stretch
and the HWC->CHW transposes of `MultimodalDetection.__getitem__` (src/datasets/MultimodalDetection.py:196-255).

The resampling itself lives in cv2 (`opencv-python`, requirements.txt), which is neither vendored in the reference nor
installed in this image.  `resize_linear` / `resize_cubic` restate OpenCV's published algorithm for floating-point images
(pixel-centre mapping src = (dst + 0.5) * src_size / dst_size - 0.5, out-of-range taps clamped to the border, bicubic
kernel A = -0.75, separable, horizontal pass first).  PARITY UNPINNED against cv2 itself; pinned only by the hand-computed
cases in tests/test_input_pipeline.py.
"""
from __future__ import annotations

import numpy as np

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406])
IMAGENET_STD = np.array([0.229, 0.224, 0.225])


def _coords(dst: int, src: int):
    f = (np.arange(dst, dtype=np.float64) + 0.5) * (src / dst) - 0.5
    s = np.floor(f).astype(np.int64)
    return s, f - s


def resize_linear(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_LINEAR) for float images [H,W(,C)]."""
    a = img.astype(np.float64)
    if a.ndim == 2:
        a = a[:, :, None]
    H, W, _ = a.shape
    sy, wy = _coords(out_h, H)
    sx, wx = _coords(out_w, W)
    wy = np.where((sy < 0) | (sy >= H - 1), 0.0, wy); sy = np.clip(sy, 0, H - 1)
    wx = np.where((sx < 0) | (sx >= W - 1), 0.0, wx); sx = np.clip(sx, 0, W - 1)
    sy1, sx1 = np.minimum(sy + 1, H - 1), np.minimum(sx + 1, W - 1)
    rows = a[:, sx] * (1 - wx)[None, :, None] + a[:, sx1] * wx[None, :, None]
    out = rows[sy] * (1 - wy)[:, None, None] + rows[sy1] * wy[:, None, None]
    return out if img.ndim == 3 else out[:, :, 0]


def _cubic_w(x: np.ndarray) -> np.ndarray:
    A = -0.75
    w0 = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A
    w1 = ((A + 2) * x - (A + 3)) * x * x + 1
    w2 = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1
    return np.stack([w0, w1, w2, 1 - w0 - w1 - w2], 0)


def resize_cubic(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_CUBIC) for float images [H,W,C]."""
    a = img.astype(np.float64)
    H, W, _ = a.shape
    sy, fy = _coords(out_h, H)
    sx, fx = _coords(out_w, W)
    wy, wx = _cubic_w(fy), _cubic_w(fx)
    rows = sum(a[:, np.clip(sx - 1 + k, 0, W - 1)] * wx[k][None, :, None] for k in range(4))
    return sum(rows[np.clip(sy - 1 + k, 0, H - 1)] * wy[k][:, None, None] for k in range(4))


def resized_hw(height: int, width: int, common_size: int):
    """Resizer: the longer side -> common_size, the other int(side * scale) (transformations.py:414-421)."""
    if height > width:
        scale = common_size / height
        return common_size, int(width * scale)
    scale = common_size / width
    return int(height * scale), common_size


def letterbox(img: np.ndarray, common_size: int) -> np.ndarray:
    """Resizer for rgb / thermal / depth: bilinear to (rh, rw), pasted top-left on a zero canvas."""
    rh, rw = resized_hw(img.shape[0], img.shape[1], common_size)
    r = resize_linear(img, rh, rw)
    out = np.zeros((common_size, common_size) + img.shape[2:], dtype=np.float64)
    out[:rh, :rw] = r
    return out


def prepare_rgb(raw_u8: np.ndarray, common_size: int) -> np.ndarray:
    """uint8 [H,W,3] (already RGB, cropped) -> float32 [3,S,S]: /255, Normalizer, Resizer, transpose."""
    x = (raw_u8.astype(np.float32) / 255.0 - IMAGENET_MEAN[None, None]) / IMAGENET_STD[None, None]
    return np.transpose(letterbox(x, common_size), (2, 0, 1)).astype(np.float32)


def prepare_depth(raw_u8: np.ndarray, common_size: int) -> np.ndarray:
    return np.transpose(letterbox(raw_u8.astype(np.float32) / 255.0, common_size), (2, 0, 1)).astype(np.float32)


def prepare_thermal(raw_u16: np.ndarray, common_size: int, ir_min: float = 20800, ir_max: float = 27000) -> np.ndarray:
    """uint16 [H,W] -> float32 [1,S,S]: clamp, cv2.normalize(NORM_MINMAX, 0..255) into uint16 (round-half-even), /255."""
    t = np.clip(raw_u16.astype(np.float64), ir_min, ir_max)
    mn, mx = t.min(), t.max()
    k = 255.0 / (mx - mn) if mx > mn else 0.0
    t = np.rint((t - mn) * k)
    return letterbox((t.astype(np.float32) / 255.0), common_size)[None].astype(np.float32)


def prepare_audio(spec: np.ndarray, common_size: int) -> np.ndarray:
    """[h,w,8] dB-mel stack -> float32 [8,S,S] (bicubic)."""
    return np.transpose(resize_cubic(spec, common_size, common_size), (2, 0, 1)).astype(np.float32)
