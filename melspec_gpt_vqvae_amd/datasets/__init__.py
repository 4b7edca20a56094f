"""On-disk formats of the reference's data pipeline: `<clip>_mel.npy` (80, 860) float spectrograms in
[0, 1], `<clip>_mel_code.npy` (5, 53) int64 VQ codes, the `data/vas_*.txt` / `data/vggsound_*.txt` split lists, and the
Lightning checkpoint key conventions.  Host-side plumbing only: numpy + torch DataLoader, no GPU work here."""
from .datamodule import DataModule
from .specs import ClipRecord, SpecCodeDataset
from .transforms import Crop
from .vas import VASSpecs
from .vggsound import VGGSoundSpecs

__all__ = ["DataModule", "Crop", "ClipRecord", "SpecCodeDataset", "VASSpecs", "VGGSoundSpecs"]
