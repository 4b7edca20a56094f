#!/usr/bin/env python3
"""Per-launch durations of ONE MelGAN generator forward (rocprofv3 --kernel-trace of this script; the last forward's
launches are listed in order with their grid sizes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from melspec_gpt_vqvae_amd.vocoder import Generator

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
g = Generator(80, 32, 3).to("cuda:0").eval()
for m in g.modules():
    object.__setattr__(m, "compute_dtype", torch.bfloat16)
x = torch.randn(B, 80, 848, device="cuda:0")
for _ in range(3):
    g(x)
torch.cuda.synchronize()
