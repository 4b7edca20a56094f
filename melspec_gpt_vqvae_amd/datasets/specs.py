"""One loader for the on-disk contract the training path consumes:

    <clip>_mel.npy        (80, 860) float spectrogram in [0, 1]     (extract_mel_spectrogram.get_spectrogram)
    <clip>_mel_code.npy   (5, 53)   int64 VQ codes, optional         (extract_codes)

A dataset is a MANIFEST - a list of ClipRecord(spec file, code file, label, target) - plus a crop rule.  What
differs between the reference's two dataset families (datasets/vas.py:30-93, datasets/vggsound.py:21-174) is only
how the manifest is derived from split lists / a label table and where the code files sit; that lives in vas.py and
vggsound.py.  Items carry the keys the Lightning modules read (minGPT.py:387-402, Lit_GPT_VAE.py:229-240):
    image (crop of 2*mel-1, the VQ-VAE's [-1, 1] range), codes (when the code file exists), target, label, file_path_.
"""
from __future__ import annotations

import os
from typing import NamedTuple, Sequence

import numpy as np
import torch

from .transforms import Crop

MEL_SUFFIX = "_mel.npy"
CODE_SUFFIX = "_mel_code.npy"
CODES_DIRNAME = "codes_10s"


class ClipRecord(NamedTuple):
    spec_path: str
    codes_path: str
    label: str
    target: int


def sibling_codes_dir(spec_dir: str) -> str:
    """`.../<something>/melspec_10s_22050hz[/]` -> `.../<something>/codes_10s` (extract_codes.py:31-35 writes there)."""
    return os.path.join(os.path.dirname(spec_dir.rstrip("/")), CODES_DIRNAME)


def class_index(labels) -> dict:
    """label -> integer target, by sorted label name (both reference families number their classes this way)."""
    return {name: k for k, name in enumerate(sorted(set(labels)))}


class SpecCodeDataset(torch.utils.data.Dataset):
    def __init__(self, records: Sequence[ClipRecord], crop_hw=(None, None), random_crop=False):
        super().__init__()
        self.records = list(records)
        self.transforms = Crop(list(crop_hw), random_crop)

    def __len__(self):
        return len(self.records)

    def __getitem__(self, i):
        rec = self.records[i]
        mel = self.transforms(np.load(rec.spec_path))
        out = {"image": 2 * mel - 1, "target": rec.target, "label": rec.label, "file_path_": rec.spec_path}
        if os.path.isfile(rec.codes_path):
            out["codes"] = np.load(rec.codes_path)
        return out
