"""Item transforms of the reference's datasets (datasets/transforms.py:14-91), without albumentations: the crop
windows are restated from albumentations-1.2.1 (requirements.txt) - CenterCrop: y1 = (H - h) // 2, x1 = (W - w) // 2
(for the (80, 860) -> (80, 848) case: columns 6 .. 853); RandomCrop: y1 = int((H - h) * random.random()), then
x1 = int((W - w) * random.random()) with Python's `random` module, in that order."""
from __future__ import annotations

import random


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
