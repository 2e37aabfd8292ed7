"""Import shim for the *reference* MM-DistillNet tree (THIS CONTAINER ONLY).

Used only by tools/oracle/make_golden.py and tools/oracle/check_oracle.py to
  (a) validate the CPU restatement in oracle/ against the reference's own code and
  (b) generate the golden vectors committed under tests/golden/.
Nothing here travels to the GPU box as reference source: it only registers stub
modules for third-party imports that are absent in this image and then imports
`/root/reference/src/...` in place (read-only tree, no bytecode written).

Stubbed third-party modules: torchvision(.ops(.boxes)), cv2,
librosa(.display), google_drive_downloader, tensorboardX, hpbandster(...),
albumentations.  The NMS stubs restate torchvision's published greedy NMS
(sort by score desc, suppress IoU > thr, IoU without the +1 convention, kept
indices in score order; batched_nms offsets boxes by idx*(max_coord+1)).
"""
import os
import sys
import types

REF = os.environ.get("MMD_REFERENCE", "/root/reference")


def _nms(boxes, scores, iou_threshold):
    import torch
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    x1, y1, x2, y2 = boxes.unbind(1)
    areas = (x2 - x1) * (y2 - y1)
    order = torch.sort(scores, descending=True, stable=True)[1].tolist()
    n = len(order)
    sup = [False] * n
    keep = []
    bx = boxes.tolist()
    ar = areas.tolist()
    import numpy as np
    f32 = np.float32
    for _i in range(n):
        i = order[_i]
        if sup[i]:
            continue
        keep.append(i)
        ix1, iy1, ix2, iy2 = (f32(v) for v in bx[i])
        ia = f32(ar[i])
        for _j in range(_i + 1, n):
            j = order[_j]
            if sup[j]:
                continue
            xx1 = max(ix1, f32(bx[j][0]))
            yy1 = max(iy1, f32(bx[j][1]))
            xx2 = min(ix2, f32(bx[j][2]))
            yy2 = min(iy2, f32(bx[j][3]))
            w = max(f32(0), f32(xx2 - xx1))
            h = max(f32(0), f32(yy2 - yy1))
            inter = f32(w * h)
            ovr = f32(inter / f32(f32(ia + f32(ar[j])) - inter))
            if ovr > f32(iou_threshold):
                sup[j] = True
    return torch.tensor(keep, dtype=torch.int64)


def _batched_nms(boxes, scores, idxs, iou_threshold):
    import torch
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + 1)
    boxes_for_nms = boxes + offsets[:, None]
    return _nms(boxes_for_nms, scores, iou_threshold)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, k):
        return _Anything()

    def __call__(self, *a, **k):
        return _Anything()


def _module_getattr(k):
    if k.startswith("__"):
        raise AttributeError(k)
    return _Anything()


def install():
    """Register stubs + put the reference on sys.path. Idempotent."""
    import torch  # noqa: F401  (import the real torch before any stub exists)
    if "torchvision" not in sys.modules:
        tv = _mod("torchvision")
        ops = _mod("torchvision.ops", nms=_nms)
        boxes = _mod("torchvision.ops.boxes", nms=_nms, batched_nms=_batched_nms)
        tv.ops = ops
        ops.boxes = boxes
        tv.transforms = _mod("torchvision.transforms", Compose=_Anything)
    for name in ["cv2", "librosa", "librosa.display", "albumentations",
                 "hpbandster", "hpbandster.core", "hpbandster.core.result",
                 "hpbandster.visualization"]:
        if name not in sys.modules:
            m = _mod(name)
            m.__getattr__ = _module_getattr  # type: ignore
    if "google_drive_downloader" not in sys.modules:
        _mod("google_drive_downloader", GoogleDriveDownloader=_Anything)
    if "tensorboardX" not in sys.modules:
        _mod("tensorboardX", SummaryWriter=_Anything)
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)


def load_train_methods():
    """Import src/optimization/train_methods.py with the numpy>=1.25 fix applied in memory.

    `batch_labels[i] == []` with an ndarray raises under numpy 2; the
    patched text uses the reference's own `isListEmpty` helper instead. The source file is
    read, patched and exec'd in memory; nothing is written anywhere.
    """
    install()
    import importlib
    import src.utils.utils  # noqa: F401  (reference module)
    path = os.path.join(REF, "src/optimization/train_methods.py")
    text = open(path).read()
    text = text.replace("batch_labels[i] == []", "isListEmpty(batch_labels[i])")
    text = text.replace("batch_labels[1] != [] and batch_labels[0] != []",
                        "(not isListEmpty(batch_labels[1])) and (not isListEmpty(batch_labels[0]))")
    mod = types.ModuleType("src.optimization.train_methods_patched")
    mod.__file__ = path
    # isListEmpty must be importable in that namespace
    from src.utils.utils import isListEmpty
    mod.__dict__["isListEmpty"] = isListEmpty
    exec(compile(text, path, "exec"), mod.__dict__)
    return mod
