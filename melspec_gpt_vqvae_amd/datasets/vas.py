"""VAS manifest builder: same constructor as the reference's VASSpecs (datasets/vas.py:30-60), items come from
specs.SpecCodeDataset.  Layout on disk:
    <root>/features/<class>/melspec_10s_22050hz/<video>_mel.npy
    <root>/features/<class>/codes_10s/<video>_mel_code.npy
    <splits_dir>/vas_{train,valid}.txt          one `<class>/<video>` per line
`spec_dir_path` carries a `*` where the class name goes, like the reference's config value
(config/config_GPT_vas.py: "./data/vas/features/*/melspec_10s_22050hz")."""
from __future__ import annotations

import os

from .specs import CODE_SUFFIX, MEL_SUFFIX, ClipRecord, SpecCodeDataset, class_index, sibling_codes_dir


def vas_manifest(lines, spec_dir_pattern):
    """split-file lines -> (records, label2target).  Targets number the classes PRESENT in `lines`, sorted by name."""
    pairs = [ln.split("/") for ln in lines]
    label2target = class_index(cls for cls, _ in pairs)
    codes_pattern = sibling_codes_dir(spec_dir_pattern)
    records = [ClipRecord(os.path.join(spec_dir_pattern.replace("*", cls), vid + MEL_SUFFIX),
                          os.path.join(codes_pattern.replace("*", cls), vid + CODE_SUFFIX), cls, label2target[cls])
               for cls, vid in pairs]
    return records, label2target


class VASSpecs(SpecCodeDataset):
    def __init__(self, split, spec_dir_path, mel_num=None, spec_len=None, spec_crop_len=None, random_crop=None,
                 crop_coord=None, for_which_class=None, splits_dir='./data'):
        self.split, self.spec_dir_path = split, spec_dir_path
        self.split_path = os.path.join(splits_dir, f"vas_{split}.txt")
        with open(self.split_path) as f:      # a missing split list is an error here (the reference prints, then fails)
            wanted = [ln for ln in f.read().splitlines() if ln and (not for_which_class or ln.startswith(for_which_class))]
        self.dataset = wanted
        records, self.label2target = vas_manifest(wanted, spec_dir_path)
        self.codes_dir_path = sibling_codes_dir(spec_dir_path)
        super().__init__(records, (mel_num, spec_crop_len), random_crop)
