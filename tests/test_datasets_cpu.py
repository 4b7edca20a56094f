"""On-disk formats + loaders on a synthetic tree shaped like BASELINE config 1: 8 classes x 2 clips,
`*_mel.npy` (80, 860) in [0, 1], `*_mel_code.npy` (5, 53) int64, split lists; checkpoint key conventions.  CPU only."""
import os
import random

import numpy as np
import pytest
import torch

import synth

CLASSES = ["baby", "cough", "dog", "drum", "fireworks", "gun", "hammer", "sneeze"]


def _make_vas_tree(root):
    lines = []
    for ci, cls in enumerate(CLASSES):
        d = os.path.join(root, "vas", "features", cls, "melspec_10s_22050hz")
        os.makedirs(d)
        for v in range(2):
            mel = synth.uniform(100 * ci + v, (80, 860), 0.0, 1.0).astype(np.float32)
            np.save(os.path.join(d, f"video_{v:05d}_mel.npy"), mel)
            lines.append(f"{cls}/video_{v:05d}")
        cd = os.path.join(root, "vas", "features", cls, "codes_10s")
        os.makedirs(cd)
        for v in range(2):
            np.save(os.path.join(cd, f"video_{v:05d}_mel_code.npy"), synth.randint(7 + ci + v, 0, 128, (5, 53)))
    sp = os.path.join(root, "splits")
    os.makedirs(sp)
    open(os.path.join(sp, "vas_train.txt"), "w").write("\n".join(lines[:12]) + "\n")
    open(os.path.join(sp, "vas_valid.txt"), "w").write("\n".join(lines[12:]) + "\n")
    return os.path.join(root, "vas", "features", "*", "melspec_10s_22050hz"), sp


def test_crop_windows_follow_albumentations_rules():
    from melspec_gpt_vqvae_amd.datasets.transforms import Crop, crop_window

    assert crop_window((80, 860), (80, 848)) == (0, 6)          # columns [6:854]
    x = np.arange(80 * 860, dtype=np.float32).reshape(80, 860)
    assert np.array_equal(Crop([80, 848], False)(x), x[:, 6:854])
    item = Crop([80, 848], False)({"input": x})
    assert item["input"].shape == (80, 848)
    random.seed(5)
    h, w = random.random(), random.random()
    random.seed(5)
    assert crop_window((90, 900), (80, 848), True) == (int(10 * h), int(52 * w))
    assert Crop([None, None])(x) is x
    with pytest.raises(ValueError):
        crop_window((80, 800), (80, 848))


def test_vas_dataset_items_and_datamodule(tmp_path):
    from melspec_gpt_vqvae_amd.datasets import DataModule, VASSpecs

    spec_dir, splits = _make_vas_tree(str(tmp_path))
    ds = VASSpecs("train", spec_dir, mel_num=80, spec_len=860, spec_crop_len=848, random_crop=False, splits_dir=splits)
    assert len(ds) == 12 and ds.label2target == {c: i for i, c in enumerate(CLASSES[:6])}
    it = ds[4]                                   # dog/video_00000
    raw = np.load(it["file_path_"])
    assert it["label"] == "dog" and it["target"] == 2 and it["image"].shape == (80, 848)
    assert np.array_equal(it["image"], 2 * raw[:, 6:854] - 1) and it["image"].min() >= -1 and it["image"].max() <= 1
    assert it["codes"].shape == (5, 53) and it["codes"].dtype == np.int64
    os.remove(os.path.join(str(tmp_path), "vas", "features", "hammer", "codes_10s", "video_00001_mel_code.npy"))
    va = VASSpecs("valid", spec_dir, 80, 860, 848, False, splits_dir=splits)
    assert "codes" in va[0] and "codes" not in va[1]          # hammer/video_00001 has no code file (valid split)
    only = VASSpecs("train", spec_dir, 80, 860, 848, False, for_which_class="dog", splits_dir=splits)
    assert len(only) == 2 and only.label2target == {"dog": 0}
    dm = DataModule(4, spec_dir, num_workers=0, mel_num=80, spec_len=860, spec_crop_len=848, random_crop=False,
                    splits_dir=splits).setup()
    batches = list(dm.train_dataloader())
    assert len(batches) == 3 and batches[0]["image"].shape == (4, 80, 848) and batches[0]["target"].dtype == torch.int64
    assert batches[0]["codes"].shape == (4, 5, 53)
    # data parallel: two ranks see disjoint halves of the (padded) permutation
    a = DataModule(2, spec_dir, 0, 80, 860, 848, False, rank=0, world=2, seed=3, splits_dir=splits).setup()
    b = DataModule(2, spec_dir, 0, 80, 860, 848, False, rank=1, world=2, seed=3, splits_dir=splits).setup()
    fa = [p for bt in a.train_dataloader() for p in bt["file_path_"]]
    fb = [p for bt in b.train_dataloader() for p in bt["file_path_"]]
    assert len(fa) == len(fb) == 6 and not set(fa) & set(fb)


def test_vggsound_layout_and_split_files(tmp_path):
    from melspec_gpt_vqvae_amd.datasets import VGGSoundSpecs

    root = str(tmp_path)
    specs = os.path.join(root, "vggsound", "melspec_10s_22050hz")
    os.makedirs(specs)
    rows, labels = [], ["dog barking", "rain"]
    for i in range(12):
        vid = f"vid{i:08d}"                       # 11 characters, like a YouTube id
        rows.append([vid, "30", labels[i % 2], "test" if i >= 8 else "train"])
        np.save(os.path.join(specs, f"{vid}_30000_40000_mel.npy"), synth.uniform(i, (80, 860), 0, 1).astype(np.float32))
    meta = os.path.join(root, "vggsound.csv")
    open(meta, "w").write("\n".join(",".join(r) for r in rows) + "\n")
    os.makedirs(os.path.join(root, "vggsound", "codes_10s"))
    np.save(os.path.join(root, "vggsound", "codes_10s", "vid00000001_30000_40000_mel_code.npy"),
            synth.randint(1, 0, 1024, (5, 53)))
    splits = os.path.join(root, "splits")
    ds = VGGSoundSpecs("train", specs + "/", mel_num=80, spec_len=860, spec_crop_len=848, random_crop=False,
                       splits_path=splits, meta_path=meta)
    tr = open(os.path.join(splits, "vggsound_train.txt")).read().split()
    va = open(os.path.join(splits, "vggsound_valid.txt")).read().split()
    te = open(os.path.join(splits, "vggsound_test.txt")).read().split()
    assert len(te) == 4 and len(va) == 4 and len(tr) == 4 and not set(tr) & set(va)     # 2 per class in test -> 2 per class in valid
    assert len(ds) == 4 and ds.class_counts.tolist() == [2, 2]
    it = ds[0]
    assert it["image"].shape == (80, 848) and it["label"] in labels and it["file_path_"].endswith("_mel.npy")
    withcodes = [ds[i] for i in range(len(ds)) if "codes" in ds[i]]
    assert all(c["codes"].shape == (5, 53) for c in withcodes)


def test_code_file_naming_and_checkpoint_key_maps(tmp_path):
    from melspec_gpt_vqvae_amd import checkpoint as ck
    from melspec_gpt_vqvae_amd.feature_extraction.extract_codes import code_path_for, list_mel_files

    p = "/d/vas/features/dog/melspec_10s_22050hz/video_00003_mel.npy"
    assert code_path_for(p) == "/d/vas/features/dog/codes_10s/video_00003_mel_code.npy"   # extract_codes.py:31-35,52
    spec_dir, _ = _make_vas_tree(str(tmp_path))
    files = list_mel_files(os.path.join(str(tmp_path), "vas", "features"))
    assert len(files) == 16 and files == sorted(files, key=lambda f: (f.split("/")[-3], f))
    lin = torch.nn.Linear(4, 3)
    wrapped = {"state_dict": {"transformer.head.weight": torch.ones(3, 4), "transformer.head.bias": torch.zeros(3),
                              "encoder.x": torch.zeros(1)}, "epoch": 3}
    torch.save(wrapped, os.path.join(str(tmp_path), "lit.ckpt"))
    ck.load_state_dict_any(lin, os.path.join(str(tmp_path), "lit.ckpt"), prefix="transformer.head")
    assert torch.equal(lin.weight, torch.ones(3, 4))
    lin2 = torch.nn.Linear(4, 3)
    ck.load_state_dict_any(lin2, wrapped)                      # prefix discovered
    assert torch.equal(lin2.weight, torch.ones(3, 4))
    with pytest.raises(RuntimeError):
        ck.load_state_dict_any(torch.nn.Linear(4, 3), {"weight": torch.ones(3, 4), "extra": torch.ones(1)})

    class VAE(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = torch.nn.Linear(2, 2)
            self.decoder = torch.nn.Linear(2, 2)

    vae, before = VAE(), None
    before = vae.decoder.weight.clone()
    res = ck.warm_start_encoder(vae, {"state_dict": {"encoder.weight": torch.full((2, 2), 7.0), "encoder.bias": torch.zeros(2),
                                                     "decoder.weight": torch.zeros(2, 2)}})
    assert torch.equal(vae.encoder.weight, torch.full((2, 2), 7.0)) and torch.equal(vae.decoder.weight, before)
    assert "decoder.weight" in res.missing_keys
