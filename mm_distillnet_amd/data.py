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
        self.seed = int(config.get('seed', 24)) + (0 if mode == "train" else 100003)
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


def collate(batch):
    items = list(zip(*batch))
    return [torch.stack(items[0]), torch.stack(items[1]), torch.stack(items[2]), torch.stack(items[3]), list(items[4]),
            list(items[5])]
