#!/usr/bin/env python3
"""VQ-VAE encoder at batch B (mel tiles -> codes), 16-bit lane: run under `rocprofv3 --kernel-trace` for the per-launch
table of one encode (tools/lab/trace_table.py prints the last encode of the trace in launch order)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from melspec_gpt_vqvae_amd import _ffi
from melspec_gpt_vqvae_amd.vqvae import big_model_attn_gan as vq

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.manual_seed(0)
m = vq.LitVQVAE(num_embeddings=128, embedding_dim=256).to("cuda:0").eval()
vq.set_compute_dtype(m, _ffi.HALF_DTYPE)
x = torch.rand(B, 1, 80, 848, device="cuda:0") * 2 - 1
with torch.no_grad():
    for _ in range(N):
        out = m.encode(x)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.no_grad():
    s.record()
    for _ in range(5):
        out = m.encode(x)
    e.record()
torch.cuda.synchronize()
print("ms per encode", s.elapsed_time(e) / 5)
