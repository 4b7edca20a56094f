#!/usr/bin/env python3
"""Per-GPU workload of BASELINE.json configs[3] (8 x MI355X DP: GPT-VAE, config_GPT_VAE_vggsound.py:43-58) on ONE GPU:
GPT-VAE XL - encoder GPT (bidirectional, last_linear = 2C) + decoder GPT, V = 1024, 40 layers, 23 heads, C = 1472,
block 265/266, dropout 0 - one training step (loss = rec + kl_weight * KL, backward, AdamW) at batch 128 per GPU, bf16.
Algorithmic FLOPs per sequence and step: 3 x (570.07 + 568.57) GFLOP.  One JSON line.
The 8-GPU run itself is the driver's (`bench.py --gpus N` on the class-GPT workload); this script reports what one rank
of configs[3] does and how many bytes its gradient exchange would move."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch

import synth
from melspec_gpt_vqvae_amd.optim import FusedAdamW
from melspec_gpt_vqvae_amd.transformer.Lit_GPT_VAE import GPT_VAE
from melspec_gpt_vqvae_amd.transformer.minGPT import set_compute_dtype

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--layers", type=int, default=40)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    torch.manual_seed(0)
    args = synth.gpt_args(vocab_size=1024, n_layer=a.layers, n_head=23, n_embd=1472, block_size=265, fix_var=0, kl_start=0.3,
                          warm_up=0, batch_size=a.batch, target_kl=0.0, beta=1.0, nsamples=1, fb=0, device=DEV,
                          learning_rate=1e-6)
    vae = GPT_VAE(args).to(DEV).train()
    set_compute_dtype(vae.encoder.transformer, torch.bfloat16)
    set_compute_dtype(vae.decoder.transformer, torch.bfloat16)
    n_params = sum(p.numel() for p in vae.parameters())
    opts = [FusedAdamW(m.transformer, lr=1e-6, betas=(0.9, 0.95), weight_decay=0.01) for m in (vae.encoder, vae.decoder)]
    x = torch.randint(0, 1024, (a.batch, 265), device=DEV)

    def step():
        total, rec, kl = vae.loss(x, 0.5, nsamples=1)
        loss = total.mean()
        for o in opts:
            o.zero_grad()
        loss.backward()
        for o in opts:
            o.step()
        return loss

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    flop_seq = 3.0 * (570.07e9 + 568.57e9) * a.layers / 40.0
    print(json.dumps({
        "bench": "GPT-VAE XL training step (BASELINE configs[3], one rank), bf16", "batch_per_gpu": a.batch,
        "layers": a.layers, "params": n_params, "ms_per_step": round(1e3 * dt, 1), "seq_per_s": round(a.batch / dt, 2),
        "algorithmic_TFLOP_per_step": round(flop_seq * a.batch / 1e12, 1),
        "TFLOPs": round(flop_seq * a.batch / dt / 1e12, 1), "frac_mfma": round(flop_seq * a.batch / dt / 2.5e15, 4),
        "grad_allreduce_bytes_fp32": 4 * n_params, "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1),
        "loss": round(float(loss), 4)}))


if __name__ == "__main__":
    main()
