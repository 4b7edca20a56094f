"""Item transforms of the reference's datasets (datasets/transforms.py:14-91), without albumentations: the crop
windows are restated from albumentations-1.2.1 (requirements.txt) - CenterCrop: y1 = (H - h) // 2, x1 = (W - w) // 2
(for the (80, 860) -> (80, 848) case: columns 6 .. 853); RandomCrop: y1 = int((H - h) * random.random()), then
x1 = int((W - w) * random.random()) with Python's `random` module, in that order."""
from __future__ import annotations

import os
import random
from pathlib import Path

import numpy as np
import torch


def crop_window(shape, crop_hw, random_crop=False):
    H, W = shape[:2]
    h, w = crop_hw
    if h > H or w > W:
        raise ValueError(f"Requested crop size ({h}, {w}) is larger than the image size ({H}, {W})")
    if random_crop:
        h_start, w_start = random.random(), random.random()
        return int((H - h) * h_start), int((W - w) * w_start)
    return (H - h) // 2, (W - w) // 2


class Crop:
    """Crop([mel_num, spec_len], random_crop) on item['input'] (dict items) or on a bare array (extract_codes.py:13-29)."""

    def __init__(self, cropped_shape=None, random_crop=False):
        self.cropped_shape = cropped_shape
        self.random_crop = bool(random_crop)
        if cropped_shape is not None and (cropped_shape[0] is None or cropped_shape[1] is None):
            self.cropped_shape = None

    def _crop(self, img):
        if self.cropped_shape is None:
            return img
        y1, x1 = crop_window(img.shape, self.cropped_shape, self.random_crop)
        return img[y1:y1 + self.cropped_shape[0], x1:x1 + self.cropped_shape[1]]

    def __call__(self, item):
        if isinstance(item, dict):
            item['input'] = self._crop(item['input'])
            return item
        return self._crop(item)


class StandardNormalizeAudio:
    """frequency-wise (x - mean_f) / std_f with the statistics file of the reference
    (data/train_means_stds_melspec_10s_22050hz.txt: 80 rows `mean std`), computed and cached when absent."""

    def __init__(self, specs_dir, train_ids_path='./data/vggsound_train.txt', cache_path='./data/'):
        self.specs_dir = specs_dir
        self.train_ids_path = train_ids_path
        self.cache_path = os.path.join(cache_path, f'train_means_stds_{Path(specs_dir).stem}.txt')
        self.train_stats = self.calculate_or_load_stats()

    def __call__(self, item):
        if isinstance(item, dict):
            key = 'input' if 'input' in item else 'image' if 'image' in item else None
            if key is None:
                raise NotImplementedError
            item[key] = (item[key] - self.train_stats['means']) / self.train_stats['stds']
        elif isinstance(item, torch.Tensor):
            item = (item - self.train_stats['means']) / self.train_stats['stds']
        else:
            raise NotImplementedError
        return item

    def calculate_or_load_stats(self):
        try:
            means, stds = np.loadtxt(self.cache_path).T
        except OSError:
            ids = [i.rstrip() for i in open(self.train_ids_path)]
            means = np.zeros((len(ids), 0))
            m, s = [], []
            for i in ids:
                spec = np.load(os.path.join(self.specs_dir, f'{i}_mel.npy'))
                m.append(spec.mean(axis=1))
                s.append(spec.std(axis=1))
            means, stds = np.array(m).mean(axis=0), np.array(s).mean(axis=0)
            np.savetxt(self.cache_path, np.vstack([means, stds]).T, fmt='%0.8f')
        return {'means': means.reshape(-1, 1), 'stds': stds.reshape(-1, 1)}


class ToTensor:
    def __call__(self, item):
        item['input'] = torch.from_numpy(item['input']).float()
        item['target'] = torch.tensor(item['target'])
        return item
