"""VGGSound spectrogram / code dataset (reference datasets/vggsound.py:18-174).  Layout on disk:
    <root>/melspec_10s_22050hz/<youtube id>_<start ms>_<end ms>_mel.npy     (80, 860) float in [0, 1]
    <root>/codes_10s/<same stem>_mel_code.npy                               (5, 53) int64
    <splits_path>/vggsound_{train,valid,test}.txt                           one clip stem per line
    <meta_path> vggsound.csv                                                rows: id, start, label, train|test
The first 11 characters of a clip stem are the YouTube id that keys the label table."""
from __future__ import annotations

import collections
import csv
import os
import random
from glob import glob
from pathlib import Path

import numpy as np
import torch

from .transforms import Crop


class VGGSound(torch.utils.data.Dataset):
    def __init__(self, split, specs_dir, transforms=None, splits_path='./data', meta_path='./data/vggsound.csv'):
        super().__init__()
        self.split, self.specs_dir, self.transforms = split, specs_dir, transforms
        self.splits_path, self.meta_path = splits_path, meta_path
        meta = list(csv.reader(open(meta_path), quotechar='"'))
        unique_classes = sorted(set(row[2] for row in meta))
        self.label2target = {label: target for target, label in enumerate(unique_classes)}
        self.target2label = {target: label for label, target in self.label2target.items()}
        self.video2target = {row[0]: self.label2target[row[2]] for row in meta}
        parts = specs_dir.split("/")[:-1]
        parts[-1] = "codes_10s"
        self.codes_dir_path = '/'.join(parts)
        self.feat_codes_suffix = '_mel_code.npy'
        split_file = os.path.join(splits_path, f'vggsound_{split}.txt')
        if not os.path.exists(split_file):
            self.make_split_files()
        ids = open(split_file).read().splitlines()
        self.dataset = [os.path.join(specs_dir, v + '_mel.npy') for v in ids]
        counts = collections.Counter(self.video2target[Path(p).stem[:11]] for p in self.dataset)
        self.class_counts = torch.tensor([counts[c] for c in range(len(counts))])

    def __getitem__(self, idx):
        spec_path = self.dataset[idx]
        video_name = Path(spec_path).stem[:11]
        codes_path = os.path.join(self.codes_dir_path,
                                  spec_path.split('/')[-1].replace('_mel.npy', self.feat_codes_suffix))
        item = {'input': np.load(spec_path), 'input_path': spec_path, 'target': self.video2target[video_name]}
        item['label'] = self.target2label[item['target']]
        if self.transforms is not None:
            item = self.transforms(item)
        if os.path.isfile(codes_path):
            item["codes"] = np.load(codes_path)
        return item

    def __len__(self):
        return len(self.dataset)

    def make_split_files(self):
        """reference :95-148: seed 1337; videos the csv marks `test` stay test; per class, as many `train` videos as
        that class has test videos are drawn (shuffle) into valid, the rest stay train; only clips present on disk are
        listed.  (The reference shuffles in set-iteration order, which depends on the interpreter's string hashing;
        here each class's videos are sorted first, so the split is reproducible.)"""
        random.seed(1337)
        available = sorted(glob(os.path.join(self.specs_dir, '*_mel.npy')))
        meta = list(csv.reader(open(self.meta_path), quotechar='"'))
        train_vids = {row[0] for row in meta if row[3] == 'train'}
        test_vids = {row[0] for row in meta if row[3] == 'test'}
        unique_classes = sorted(set(row[2] for row in meta))
        label2target = {label: target for target, label in enumerate(unique_classes)}
        video2target = {row[0]: label2target[row[2]] for row in meta}
        test_count = collections.Counter(video2target[v] for v in test_vids)
        train_wo_valid, valid_vids = set(), set()
        for target in range(len(unique_classes)):
            vids = sorted(v for v in train_vids if video2target[v] == target)
            random.shuffle(vids)
            valid_vids.update(vids[:test_count[target]])
            train_wo_valid.update(vids[test_count[target]:])
        os.makedirs(self.splits_path, exist_ok=True)
        files = {n: open(os.path.join(self.splits_path, f'vggsound_{n}.txt'), 'w') for n in ('train', 'valid', 'test')}
        try:
            for path in available:
                name = Path(path.replace('_mel.npy', '')).name
                vid = name[:11]
                if vid in train_wo_valid:
                    files['train'].write(name + '\n')
                elif vid in valid_vids:
                    files['valid'].write(name + '\n')
                elif vid in test_vids:
                    files['test'].write(name + '\n')
                else:
                    raise Exception(f'Clip {name} is neither in train, valid nor test. Strange.')
        finally:
            for f in files.values():
                f.close()


class VGGSoundSpecs(VGGSound):
    """VGGSound items in the VQ-VAE's convention: `image` in [-1, 1], `file_path_` (reference :151-174)."""

    def __init__(self, split, spec_dir_path, mel_num=None, spec_len=None, spec_crop_len=None, random_crop=None,
                 crop_coord=None, for_which_class=None, splits_path='./data', meta_path='./data/vggsound.csv'):
        super().__init__(split, spec_dir_path, splits_path=splits_path, meta_path=meta_path)
        if for_which_class:
            raise NotImplementedError
        self.transforms = Crop([mel_num, spec_crop_len], random_crop)

    def __getitem__(self, idx):
        item = super().__getitem__(idx)
        item['image'] = 2 * item['input'] - 1
        item['file_path_'] = item['input_path']
        item.pop('input')
        item.pop('input_path')
        return item
