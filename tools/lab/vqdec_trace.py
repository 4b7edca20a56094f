#!/usr/bin/env python3
"""VQ-VAE decoder at batch B (codes -> mel tile), 16-bit lane: kernel table via rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from melspec_gpt_vqvae_amd import _ffi
from melspec_gpt_vqvae_amd.vqvae import big_model_attn_gan as vq

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
m = vq.LitVQVAE(num_embeddings=128, embedding_dim=256).to("cuda:0").eval()
vq.set_compute_dtype(m, _ffi.HALF_DTYPE)
codes = torch.randint(0, 128, (B, 5, 53), device="cuda:0")
with torch.no_grad():
    for _ in range(3):
        y = m.decode(m._vq_vae.get_codebook_entry(codes.reshape(-1), shape=(B, 5, 53, 256)))
torch.cuda.synchronize()
print(tuple(y.shape))
