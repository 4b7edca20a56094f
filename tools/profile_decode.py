#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats`: greedy sampling of 265 tokens at batch B (argv[1], default 1),
class-GPT VAS bf16, graph replay - per-kernel durations of a decode step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import time

import torch

import synth
from melspec_gpt_vqvae_amd.transformer.minGPT import Lit_minGPT, set_compute_dtype

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
args = synth.gpt_args(n_layer=24, n_head=16, n_embd=1024, reconstruct_spec="", device="cuda:0", batch_size=2, learning_rate=1e-6)
lit = Lit_minGPT(args).to("cuda:0").eval()
set_compute_dtype(lit.transformer, torch.bfloat16)
c = torch.randint(0, 8, (B, 1), device="cuda:0")
x0 = torch.zeros(B, 0, dtype=torch.int64, device="cuda:0")
lit.sample(x0, c, steps=8, sample=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
lit.sample(x0, c, steps=265, sample=False)
torch.cuda.synchronize()
print("ms_per_token", round((time.perf_counter() - t0) / 265 * 1e3, 3))
