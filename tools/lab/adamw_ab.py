#!/usr/bin/env python3
"""LAB: fused AdamW over the class-GPT's 302.85 M parameters (30 bytes per parameter), ms per launch and TB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from melspec_gpt_vqvae_amd import ops
n = 302_854_144
p = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda") * 1e-3; m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
pb = torch.empty(n, dtype=torch.bfloat16, device="cuda")
def run(): ops.adamw(p, g, m, v, lr=1e-6, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01, step=3, param_bf16=pb)
for _ in range(3): run()
torch.cuda.synchronize()
ts = []
for _ in range(9):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): run()
    e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 5)
ts.sort(); t = ts[len(ts) // 2]
print(f"grid cap {os.environ.get('MELGPT_ADAMW_GRID', '4096')}: {t:.3f} ms  {30.0 * n / t / 1e9:.2f} TB/s")
