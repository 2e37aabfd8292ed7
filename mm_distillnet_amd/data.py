"""Input contract of the reference dataset, synthetic implementation.

`MultimodalDetection.__getitem__` (src/datasets/MultimodalDetection.py:176-257) yields
(rgb[3,S,S] ImageNet-normalised, thermal[1,S,S] /255, depth[3,S,S] /255, audio[8,S,S] dB-mel resized to SxS,
label, id).  The real corpus and its cv2/librosa pipeline are outside the hot path and not available offline;
this dataset produces tensors with the same shapes / statistics so train.py and evaluate.py run end to end."""
from __future__ import annotations

import torch
from torch.utils.data import Dataset

CLASSES = ['aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable',
           'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor']


def valid_classes_dict(valid_labels=("car",)):
    d = {'labels_i2txt': {}, 'labels_txt2i': {}, 'predictions_txt2i': {}, 'predictions_i2txt': {}}
    for i, c in enumerate(CLASSES):
        if valid_labels and c not in valid_labels:
            continue
        d['labels_txt2i'][c] = i; d['labels_i2txt'][i] = c
        d['predictions_txt2i'][c] = i; d['predictions_i2txt'][i] = c     # VOC ids == indices (BaseDataset.py:141-165)
    return d


class SyntheticMultimodalDetection(Dataset):
    def __init__(self, config, mode: str = "train", length: int = 64):
        self.size = int(config['image_size'])
        self.length = int(config.get('synthetic_length', length))
        self.seed = int(config.get('seed', 24)) + {"train": 0, "val": 50021}.get(mode, 100003)
        self.classes = CLASSES
        vl = config.get('valid_labels', None)
        self.valid_classes_dict = valid_classes_dict(tuple(vl.split(',')) if vl else None)

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        S = self.size
        rgb = torch.randn(3, S, S, generator=g)
        thermal = torch.rand(1, S, S, generator=g)
        depth = torch.rand(3, S, S, generator=g)
        raw = torch.randn(1, 8, 128, 128, generator=g) * 15.0 - 40.0
        audio = torch.nn.functional.interpolate(raw, size=(S, S), mode="bicubic", align_corners=False)[0]
        return rgb, thermal, depth, audio, None, i

    def yield_batch(self, batch_size, ids):
        return _yield_batch(self, batch_size, ids)


def _yield_batch(ds, batch_size, ids):
    """`MultimodalDetection.yield_batch` (src/datasets/MultimodalDetection.py:352-367): for every sample of the batch draw ANOTHER recording
    (numpy's global RNG, like upstream), return its RGB frame and the spectrograms of the two recordings' MIXED audio.  Upstream mixes
    the waveforms with librosa (`merge_audios`, :329-350: (a1 + a2) / 2 -> mel spectrogram -> cv2 resize); the synthetic dataset has no
    waveforms, so it mixes the dB-scale spectrograms as incoherent sources at half amplitude each: 10 log10(10^(a/10) + 10^(b/10)) - 6 dB.
    -> (rgb [B,3,S,S], audio [B,8,S,S])"""
    import numpy as np
    mine = set(int(i) for i in ids)
    pool = [i for i in range(len(ds)) if i not in mine]
    picks = np.random.choice(pool, size=batch_size)
    rgbs, audios = [], []
    for k in range(batch_size):
        rgb2, _, _, a2, _, _ = ds[int(picks[k])]
        a1 = ds[int(ids[k])][3]
        mix = 10.0 * torch.log10(torch.pow(10.0, a1 / 10.0) + torch.pow(10.0, a2 / 10.0)) - 6.0206
        rgbs.append(rgb2); audios.append(mix)
    return torch.stack(rgbs), torch.stack(audios)


def collate(batch):
    items = list(zip(*batch))
    return [torch.stack(items[0]), torch.stack(items[1]), torch.stack(items[2]), torch.stack(items[3]), list(items[4]),
            list(items[5])]


class RawSyntheticMultimodalDetection(Dataset):
    """Raw-format samples as the reference decodes them BEFORE its transforms: rgb uint8 [H,W,3], thermal uint16 [H,W]
    (sensor counts), depth uint8 [H,W,3], audio float32 [h,w,8] (8 dB-mel spectrograms stacked, MultimodalDetection.py:223-227).
    Feeds DeviceInputPipeline."""

    def __init__(self, config, mode: str = "train", length: int = 64, frame_hw=(270, 360), mel_hw=(128, 128)):
        self.length = int(config.get('synthetic_length', length))
        self.seed = int(config.get('seed', 24)) + {"train": 0, "val": 50021}.get(mode, 100003)
        self.frame_hw, self.mel_hw = frame_hw, mel_hw

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        H, W = self.frame_hw
        rgb = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8)
        thermal = torch.randint(19000, 29000, (H, W), generator=g, dtype=torch.int32).to(torch.int16)   # uint16 bit pattern
        depth = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8)
        audio = torch.randn(self.mel_hw[0], self.mel_hw[1], 8, generator=g) * 15.0 - 40.0
        return {"rgb": rgb, "thermal": thermal, "depth": depth, "audio": audio, "id": i}


RAW_KEYS = ("rgb", "thermal", "depth", "audio")


def collate_raw(batch):
    """Raw samples of one batch -> ONE dict of stacked tensors {"rgb": [B,H,W,3], "thermal": [B,H,W], "depth": [B,H,W,3], "audio": [B,h,w,8],
    "id": [...]} when every frame of the batch has the same size (the corpus' fixed camera / mel geometry): 4 shared-memory segments,
    pinned copies and H2D copies per batch instead of 4 per sample.  Mixed sizes: the list of per-sample dicts, unchanged."""
    if all(s[k].shape == batch[0][k].shape for s in batch for k in RAW_KEYS):
        out = {k: torch.stack([s[k] for s in batch]) for k in RAW_KEYS}
        out["id"] = [s["id"] for s in batch]
        return out
    return batch


class DeviceInputPipeline:
    """The reference's Normalizer + Resizer + HWC->CHW (+ thermal clamp/stretch) on the GPU (csrc/input.hip): raw frames are
    staged in pinned memory, copied on a dedicated stream and transformed there, so H2D and preprocessing of batch n+1
    overlap the step of batch n; `wait()` orders the compute stream after the batch is ready.  Replaces the per-sample
    cv2 work of `MultimodalDetection.__getitem__` + `Resizer` (src/datasets/transformations.py:407-467)."""

    MEAN = (0.485, 0.456, 0.406)
    STD = (0.229, 0.224, 0.225)

    def __init__(self, image_size: int, device, ir_min: float = 20800.0, ir_max: float = 27000.0):
        from . import _lib
        self.call = _lib.call
        self.S = int(image_size)
        self.device = torch.device(device)
        self.ir = (float(ir_min), float(ir_max))
        self.stream = torch.cuda.Stream(device=self.device)
        self.mean = torch.tensor(self.MEAN, dtype=torch.float32, device=self.device)
        self.std = torch.tensor(self.STD, dtype=torch.float32, device=self.device)
        self._pinned = {}
        self._held = []
        self._event = None
        self._batch = None

    def _stage(self, key, t: torch.Tensor) -> torch.Tensor:
        if t.is_pinned():
            # already page-locked (DataLoader(pin_memory=True) pins the batch on its own thread): copy from it directly and keep it
            # alive until the copy has been consumed (the next submit's event wait)
            self._held.append(t)
            return t.to(self.device, non_blocking=True)
        p = self._pinned.get(key)
        if p is None or p.shape != t.shape or p.dtype != t.dtype:
            p = torch.empty(t.shape, dtype=t.dtype).pin_memory()
            self._pinned[key] = p
        p.copy_(t)         # (a host copy on the calling thread: ~0.6 ms per 0.5 MB frame - 20 ms for a batch of 8 x 4 pageable tensors)
        return p.to(self.device, non_blocking=True)

    def submit(self, samples):
        """samples: list of per-sample dicts from RawSyntheticMultimodalDetection (or a real decoder with the same raw formats), or the
        stacked dict `collate_raw` makes of them."""
        stacked = isinstance(samples, dict)
        B, S, call = (samples["rgb"].shape[0] if stacked else len(samples)), self.S, self.call
        if self._event is not None:
            self._event.synchronize()          # the pinned buffers of the previous submit have been consumed
        self._held = []
        with torch.cuda.stream(self.stream):
            # allocated ON the copy stream: the caching allocator then owns these blocks for that stream, and wait()'s
            # record_stream(compute stream) defers their reuse until the step that reads them has finished (a block handed back
            # by the compute stream's pool could still be read by a queued replay while this stream overwrites it)
            out = {"rgb": torch.empty(B, 3, S, S, device=self.device), "thermal": torch.empty(B, 1, S, S, device=self.device),
                   "depth": torch.empty(B, 3, S, S, device=self.device), "audio": torch.empty(B, 8, S, S, device=self.device)}
            mm = torch.empty(B, 2, device=self.device)
            if stacked:
                whole = {k: self._stage((k, "batch"), samples[k]) for k in RAW_KEYS}
            for b in range(B):
                if stacked:
                    smp, stage = {k: whole[k][b] for k in RAW_KEYS}, (lambda key, t: t)
                else:
                    smp, stage = samples[b], self._stage
                rgb = stage(("rgb", b), smp["rgb"]); H, W = rgb.shape[:2]
                call("mmd_image_letterbox", rgb, 0, H, W, 3, 1.0 / 255.0, self.mean, self.std, 0, 0.0, 0.0, None, S, out["rgb"][b])
                d = stage(("depth", b), smp["depth"]); H, W = d.shape[:2]
                call("mmd_image_letterbox", d, 0, H, W, 3, 1.0 / 255.0, None, None, 0, 0.0, 0.0, None, S, out["depth"][b])
                t = stage(("thermal", b), smp["thermal"]); H, W = t.shape[:2]
                call("mmd_image_minmax", t, 1, H * W, self.ir[0], self.ir[1], mm[b])
                call("mmd_image_letterbox", t, 1, H, W, 1, 1.0 / 255.0, None, None, 1, self.ir[0], self.ir[1], mm[b], S,
                     out["thermal"][b])
                a = stage(("audio", b), smp["audio"]); h, w, c = a.shape
                call("mmd_resize_cubic", a, h, w, c, S, out["audio"][b])
            self._event = self.stream.record_event()
        self._batch = out
        return self

    def wait(self):
        """-> dict of [B,C,S,S] float32 device tensors, ordered before later work on the current stream."""
        torch.cuda.current_stream(self.device).wait_event(self._event)
        for t in self._batch.values():
            t.record_stream(torch.cuda.current_stream(self.device))
        return self._batch


class TensorInputPipeline:
    """Ready-made tensors (the reference loader's contract, MultimodalDetection.__getitem__: rgb / thermal / depth / audio fp32 [B,C,S,S],
    15.7 MB per sample at 512^2) to the device on a COPY stream with the same submit() / wait() interface as DeviceInputPipeline, so
    train.py keeps one batch of look-ahead on either path: the 126 MB of H2D copies of batch n+1 overlap the step of batch n instead of
    sitting in front of it on the compute stream (2.5 ms of a 14 ms step at PCIe Gen5 rates)."""
    KEYS = ("rgb", "thermal", "depth", "audio")

    def __init__(self, device):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self._event, self._batch, self._held = None, None, []

    def submit(self, item):
        if self._event is not None:
            self._event.synchronize()          # the previous submit's (pinned) sources have been read
        self._held = [item[0], item[1], item[2], item[3]]
        with torch.cuda.stream(self.stream):
            # allocated on the copy stream; wait()'s record_stream defers their reuse until the compute stream has consumed them
            out = {k: t.to(self.device, non_blocking=True) for k, t in zip(self.KEYS, self._held)}
            self._event = self.stream.record_event()
        self._batch = out
        return self

    def wait(self):
        torch.cuda.current_stream(self.device).wait_event(self._event)
        for t in self._batch.values():
            t.record_stream(torch.cuda.current_stream(self.device))
        return self._batch


def _pin(obj):
    if isinstance(obj, torch.Tensor):
        return obj if obj.is_pinned() else obj.pin_memory()
    if isinstance(obj, dict):
        return {k: _pin(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_pin(v) for v in obj)
    return obj


class CachedBatches:
    """cfg `synthetic_cache = N` (extension key, absent upstream -> off): the first N batches of `loader` are generated ONCE, kept in pinned
    host memory and cycled for the loader's length, so that what train.py's loop measures is the loop (H2D, input transforms, graph replay,
    logging), not the synthetic generator (7 x 512^2 normal draws + a bicubic resize per sample on the tensor path).  The batches still
    cross PCIe every step.  len() = the wrapped loader's, so epoch / iteration accounting is unchanged."""

    def __init__(self, loader, n: int):
        self.n_iter = len(loader)
        self.batches = []
        for i, b in enumerate(loader):
            self.batches.append(_pin(b))
            if i + 1 >= n:
                break

    def __len__(self):
        return self.n_iter

    def __iter__(self):
        for i in range(self.n_iter):
            yield self.batches[i % len(self.batches)]
