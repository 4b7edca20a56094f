#!/usr/bin/env python3
"""Lab: VQ-VAE encoder forward (64 mel tiles, 16-bit lane) timed with events, 20 repetitions - for A/B runs of two library
builds (MELGPT_LAB_LIB=... python tools/lab/enc_ab.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from melspec_gpt_vqvae_amd.vqvae import big_model_attn_gan as vq

torch.manual_seed(0)
m = vq.LitVQVAE(num_embeddings=128, embedding_dim=256).to("cuda").eval()
vq.set_compute_dtype(m, torch.bfloat16)
x = torch.randn(64, 1, 80, 848, device="cuda")
with torch.no_grad():
    for _ in range(3):
        m.encode_to_codes(x)
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); m.encode_to_codes(x); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
ts.sort()
print(f"encode 64 tiles: median {ts[10]:.3f} ms  min {ts[0]:.3f}  ({os.environ.get('MELGPT_LAB_LIB', 'in-tree')})")
