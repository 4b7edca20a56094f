"""`*_mel.npy` -> `*_mel_code.npy`: the reference's feature_extraction/extract_codes.py (:31-56, :58-118) on the HIP
VQ-VAE encoder, batched.  Per clip the reference does: np.load -> float32 -> CenterCrop(80, 848) -> 2x - 1 ->
LitVQVAE.encode -> VectorQuantizer -> indices.reshape(5, 53) -> np.save(<dir of the mel folder>/codes_10s/
<stem>_code.npy); files that already exist are skipped, unreadable ones are reported as damaged.  Here clips are
stacked into batches for LitVQVAE.encode_to_codes (one encoder pass + one fused VQ lookup per batch)."""
from __future__ import annotations

import argparse
import os
from glob import glob

import numpy as np
import torch

from ..datasets.transforms import Crop


def code_path_for(mel_path, folder_name='codes_10s'):
    save_dir = os.path.dirname(os.path.dirname(mel_path))
    audio_name = os.path.basename(mel_path).split('.')[0]
    return os.path.join(save_dir, folder_name, audio_name + '_code.npy')


@torch.no_grad()
def get_codes_batch(mel_paths, device, model, transforms, folder_name='codes_10s'):
    """encode the clips of `mel_paths` that have no code file yet; returns the list of files written."""
    todo, mels = [], []
    for p in mel_paths:
        if os.path.isfile(code_path_for(p, folder_name)):
            continue
        try:
            mel = transforms(np.load(p).astype(np.float32))
            todo.append(p)
            mels.append(2 * mel - 1)
        except Exception:
            print(p, "is damaged")
    if not todo:
        return []
    x = torch.from_numpy(np.stack(mels)).unsqueeze(1).to(device)
    codes = model.encode_to_codes(x).cpu().numpy()          # (n, 5, 53) int64
    written = []
    for p, c in zip(todo, codes):
        out = code_path_for(p, folder_name)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        np.save(out, c)
        written.append(out)
    return written


def list_mel_files(input_dir):
    """the reference's directory walk (:84-118): VGGSound keeps all clips in <input_dir>/melspec_10s_22050hz, VAS has
    one <class>/melspec_10s_22050hz folder per class."""
    if "vggsound" in input_dir:
        return sorted(glob(os.path.join(input_dir, "melspec_10s_22050hz", "*.npy")))
    paths = []
    for folder in sorted(os.listdir(input_dir)):
        paths += sorted(glob(os.path.join(input_dir, folder, "melspec_10s_22050hz", "*.npy")))
    return paths


def extract_all(input_dir, model, device="cuda", spec_crop_len=848, batch_size=64, folder_name='codes_10s'):
    transforms = Crop([80, spec_crop_len], False)
    paths = list_mel_files(input_dir)
    written = []
    for i in range(0, len(paths), batch_size):
        written += get_codes_batch(paths[i:i + batch_size], device, model, transforms, folder_name)
    return written


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-i", "--input_dir", default="data/vas/features")
    ap.add_argument("-m", "--model_dir", default="lightning_logs/2021-06-06T19-42-53_vas_codebook.pt")
    ap.add_argument("-emb_dim", "--embedding_dim", type=int, default=256)
    ap.add_argument("-n_e", "--num_embeddings", type=int, default=128)
    ap.add_argument("-crop", "--spec_crop_len", type=int, default=848)
    ap.add_argument("-b", "--batch_size", type=int, default=64)
    a = ap.parse_args(argv)
    from ..checkpoint import load_state_dict_any
    from ..vqvae.big_model_attn_gan import LitVQVAE

    model = LitVQVAE(a.num_embeddings, a.embedding_dim)
    load_state_dict_any(model, a.model_dir)
    model.eval().to("cuda")
    n = len(extract_all(a.input_dir, model, "cuda", a.spec_crop_len, a.batch_size))
    print(f"wrote {n} code files")


if __name__ == '__main__':
    main()
