"""VAS spectrogram / code dataset (reference datasets/vas.py:30-93).  Layout on disk:
    <root>/features/<class>/melspec_10s_22050hz/<video>_mel.npy        (80, 860) float, values in [0, 1]
    <root>/features/<class>/codes_10s/<video>_mel_code.npy             (5, 53) int64 (written by extract_codes)
    <splits_dir>/vas_{train,valid}.txt                                 lines `<class>/<video>`
`spec_dir_path` carries a `*` where the class name goes, exactly like the reference's config value."""
from __future__ import annotations

import os

import numpy as np
import torch

from .transforms import Crop


class VASSpecs(torch.utils.data.Dataset):
    def __init__(self, split, spec_dir_path, mel_num=None, spec_len=None, spec_crop_len=None, random_crop=None,
                 crop_coord=None, for_which_class=None, splits_dir='./data'):
        super().__init__()
        self.split = split
        self.spec_dir_path = spec_dir_path
        parts = spec_dir_path.split("/")
        parts[-1] = "codes_10s"
        self.codes_dir_path = '/'.join(parts)
        self.split_path = os.path.join(splits_dir, f'vas_{split}.txt')
        self.feat_suffix = '_mel.npy'
        self.feat_codes_suffix = '_mel_code.npy'
        if not os.path.exists(self.split_path):
            print(f'split does not exist in {self.split_path}..')
        full_dataset = open(self.split_path).read().splitlines()
        self.dataset = [v for v in full_dataset if v.startswith(for_which_class)] if for_which_class else full_dataset
        unique_classes = sorted(set(cls_vid.split('/')[0] for cls_vid in self.dataset))
        self.label2target = {label: target for target, label in enumerate(unique_classes)}
        self.transforms = Crop([mel_num, spec_crop_len], random_crop)

    def __getitem__(self, idx):
        cls, vid = self.dataset[idx].split('/')
        spec_path = os.path.join(self.spec_dir_path.replace('*', cls), f'{vid}{self.feat_suffix}')
        codes_path = os.path.join(self.codes_dir_path.replace('*', cls), f'{vid}{self.feat_codes_suffix}')
        item = {'input': np.load(spec_path), 'file_path_': spec_path, 'label': cls, 'target': self.label2target[cls]}
        if self.transforms is not None:
            item = self.transforms(item)
        item['image'] = 2 * item['input'] - 1      # the VQ-VAE expects [-1, 1]; the files hold [0, 1]
        item.pop('input')
        if os.path.isfile(codes_path):
            item["codes"] = np.load(codes_path)
        return item

    def __len__(self):
        return len(self.dataset)
