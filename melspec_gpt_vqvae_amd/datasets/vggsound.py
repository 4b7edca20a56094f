"""VGGSound manifest builder: same constructor as the reference's VGGSoundSpecs (datasets/vggsound.py:151-174), items
come from specs.SpecCodeDataset.  Layout on disk:
    <root>/melspec_10s_22050hz/<youtube id>_<start ms>_<end ms>_mel.npy
    <root>/codes_10s/<same stem>_mel_code.npy
    <splits_path>/vggsound_{train,valid,test}.txt     one clip stem per line
    <meta_path>  vggsound.csv                         rows: youtube id, start second, label, train|test
A clip's label is looked up by the first 11 characters of its stem (the YouTube id)."""
from __future__ import annotations

import csv
import os
import random
from collections import Counter, defaultdict

import torch

from .specs import CODE_SUFFIX, MEL_SUFFIX, ClipRecord, SpecCodeDataset, class_index, sibling_codes_dir

YT_ID_LEN = 11


class LabelTable:
    """vggsound.csv: which label (and official train/test part) a YouTube id has."""

    def __init__(self, meta_path):
        with open(meta_path, newline="") as f:
            rows = [r for r in csv.reader(f, quotechar='"') if r]
        self.label2target = class_index(r[2] for r in rows)
        self.target2label = {t: name for name, t in self.label2target.items()}
        self.video2target = {r[0]: self.label2target[r[2]] for r in rows}
        self.part = {r[0]: r[3] for r in rows}


def write_vggsound_splits(specs_dir, table: LabelTable, splits_path, seed=1337):
    """Derive vggsound_{train,valid,test}.txt when they are absent (the reference does this on first use,
    datasets/vggsound.py:95-148): `test` ids stay test; from every class's `train` ids as many as that class has test
    ids go to valid (seeded shuffle of the SORTED id list - reproducible, unlike the reference's set-order shuffle);
    only clips present on disk are listed; a clip whose id the table does not know is an error."""
    by_class = defaultdict(lambda: {"train": [], "test": []})
    for vid, part in table.part.items():
        by_class[table.video2target[vid]][part].append(vid)
    rng = random.Random(seed)
    bucket = {}
    for target in sorted(by_class):
        pool = sorted(by_class[target]["train"])
        rng.shuffle(pool)
        n_valid = len(by_class[target]["test"])
        bucket.update({v: "valid" for v in pool[:n_valid]})
        bucket.update({v: "train" for v in pool[n_valid:]})
        bucket.update({v: "test" for v in by_class[target]["test"]})
    listing = {"train": [], "valid": [], "test": []}
    for fn in sorted(os.listdir(specs_dir)):
        if not fn.endswith(MEL_SUFFIX):
            continue
        stem = fn[:-len(MEL_SUFFIX)]
        where = bucket.get(stem[:YT_ID_LEN])
        if where is None:
            raise KeyError(f"clip {stem}: its id is not in the label table, cannot assign it to a split")
        listing[where].append(stem)
    os.makedirs(splits_path, exist_ok=True)
    for name, stems in listing.items():
        with open(os.path.join(splits_path, f"vggsound_{name}.txt"), "w") as f:
            f.writelines(s + "\n" for s in stems)


class VGGSoundSpecs(SpecCodeDataset):
    def __init__(self, split, spec_dir_path, mel_num=None, spec_len=None, spec_crop_len=None, random_crop=None,
                 crop_coord=None, for_which_class=None, splits_path='./data', meta_path='./data/vggsound.csv'):
        if for_which_class:
            raise NotImplementedError("per-class VGGSound subsets do not exist in the reference either (:160-161)")
        self.split, self.specs_dir = split, spec_dir_path
        table = LabelTable(meta_path)
        self.label2target, self.target2label, self.video2target = table.label2target, table.target2label, table.video2target
        split_file = os.path.join(splits_path, f"vggsound_{split}.txt")
        if not os.path.exists(split_file):
            write_vggsound_splits(spec_dir_path, table, splits_path)
        with open(split_file) as f:
            stems = [ln for ln in f.read().splitlines() if ln]
        self.codes_dir_path = sibling_codes_dir(spec_dir_path)
        records = []
        for stem in stems:
            target = table.video2target[stem[:YT_ID_LEN]]
            records.append(ClipRecord(os.path.join(spec_dir_path, stem + MEL_SUFFIX),
                                      os.path.join(self.codes_dir_path, stem + CODE_SUFFIX),
                                      table.target2label[target], target))
        self.dataset = [r.spec_path for r in records]
        seen = Counter(r.target for r in records)
        self.class_counts = torch.tensor([seen[t] for t in sorted(seen)])
        super().__init__(records, (mel_num, spec_crop_len), random_crop)
