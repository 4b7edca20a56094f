"""The reference's DataModule (datasets/datamodule.py:10-90) without Lightning: picks the dataset family from the
path ("vggsound" / "vas"), builds train / valid (/ test) sets and loaders with drop_last=True, shuffle on train.
Under data parallelism pass `rank` / `world`: the loaders then use the DistributedSampler partition rule that
Lightning's DDP gives the reference (melspec_gpt_vqvae_amd.dp.distributed_shard)."""
from __future__ import annotations

import numpy as np
from torch.utils.data import DataLoader, Subset

from .vas import VASSpecs
from .vggsound import VGGSoundSpecs


class DataModule:
    def __init__(self, batch_size, spec_dir_path, num_workers=None, mel_num=None, spec_len=None, spec_crop_len=None,
                 random_crop=None, rank=0, world=1, seed=0, **dataset_kw):
        self.batch_size = batch_size
        self.num_workers = num_workers if num_workers is not None else batch_size * 2
        self.spec_dir_path = spec_dir_path
        self.kw = dict(mel_num=mel_num, spec_len=spec_len, spec_crop_len=spec_crop_len, random_crop=random_crop,
                       **dataset_kw)
        self.rank, self.world, self.seed, self.epoch = rank, world, seed, 0
        self.train_dataset = self.val_dataset = self.test_dataset = None

    def setup(self, stage=None):
        if "vggsound" in self.spec_dir_path:
            cls, splits = VGGSoundSpecs, ('train', 'valid', 'test')
        elif "vas" in self.spec_dir_path:
            cls, splits = VASSpecs, ('train', 'valid')
        else:
            raise ValueError(f"cannot tell the dataset family from {self.spec_dir_path!r} (expects 'vas' or 'vggsound')")
        sets = [cls(s, spec_dir_path=self.spec_dir_path, **self.kw) for s in splits]
        self.train_dataset, self.val_dataset = sets[0], sets[1]
        self.test_dataset = sets[2] if len(sets) > 2 else None
        return self

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _loader(self, ds, shuffle):
        if self.world > 1:
            from ..dp import distributed_shard

            # Lightning injects DistributedSampler(drop_last=False): the permutation is PADDED to a multiple of world;
            # the DataLoader's own drop_last=True (reference datamodule.py:70-72) then drops the ragged last batch
            idx = distributed_shard(len(ds), self.rank, self.world, seed=self.seed, epoch=self.epoch, shuffle=shuffle,
                                    drop_last=False)
            ds, shuffle = Subset(ds, idx), False
        return DataLoader(ds, batch_size=self.batch_size, num_workers=self.num_workers,
                          worker_init_fn=self.worker_init_fn, drop_last=True, shuffle=shuffle)

    def train_dataloader(self):
        return self._loader(self.train_dataset, True)

    def val_dataloader(self):
        return self._loader(self.val_dataset, False)

    def val_dataloader_shuffled(self):
        return self._loader(self.val_dataset, True)

    def test_dataloader(self):
        return self._loader(self.test_dataset, False)

    @staticmethod
    def worker_init_fn(worker_id):
        np.random.seed(np.random.get_state()[1][0] + worker_id)
