#!/usr/bin/env python3
"""Autoregressive sampling throughput of the class-GPT VAS model (24 L, 1024, 16 H, 266 positions, bf16):
KV-cached decode steps (GPT.decode_step) vs the reference's full re-forward per token
(transformer/minGPT.py:293-360).  One JSON line per batch size; roofline = bf16 weight bytes streamed per decode
step / step time vs 8 TB/s (a decode step at these batch sizes is bound by reading the 302.6 M weights once)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch

import synth
from melspec_gpt_vqvae_amd.transformer.minGPT import Lit_minGPT, set_compute_dtype

DEV = "cuda:0"


def main():
    args = synth.gpt_args(n_layer=24, n_head=16, n_embd=1024, reconstruct_spec="", device=DEV, batch_size=2,
                          learning_rate=1e-6)  # config/config_GPT_vas.py:1-18
    lit = Lit_minGPT(args).to(DEV).eval()
    set_compute_dtype(lit.transformer, torch.bfloat16)
    n_lin = sum(p.numel() for n, p in lit.transformer.named_parameters() if p.dim() == 2 and "emb" not in n)
    for B in (1, 16, 128):
        c = torch.randint(0, 8, (B, 1), device=DEV)
        x0 = torch.zeros(B, 0, dtype=torch.int64, device=DEV)
        lit.sample(x0, c, steps=8, sample=True, top_k=64)  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        xs, _ = lit.sample(x0, c, steps=265, sample=True, top_k=64)
        torch.cuda.synchronize()
        dt_kv = time.perf_counter() - t0
        n_ref = 24
        t0 = time.perf_counter()
        lit.sample(x0, c, steps=n_ref, sample=True, top_k=64, kv_cache=False)
        torch.cuda.synchronize()
        # the re-forward cost grows with the prefix: time the LAST 24 of 265 steps too and integrate linearly
        t1 = time.perf_counter()
        lit.sample(xs[:, :240], c, steps=n_ref, sample=True, top_k=64, kv_cache=False)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        per_short, per_long = (t1 - t0) / n_ref, (t2 - t1) / n_ref
        dt_ref = 265 * 0.5 * (per_short + per_long)
        step_ms = 1e3 * dt_kv / 265
        print(json.dumps({
            "bench": "sampling 265 tokens, class-GPT VAS bf16", "batch": B,
            "kv_cached": {"seconds": round(dt_kv, 4), "tokens_per_s": round(B * 265 / dt_kv, 1), "ms_per_step": round(step_ms, 3)},
            "reforward_estimate": {"seconds": round(dt_ref, 3), "tokens_per_s": round(B * 265 / dt_ref, 1),
                                   "ms_per_step_first24": round(1e3 * per_short, 3), "ms_per_step_last24": round(1e3 * per_long, 3)},
            "speedup": round(dt_ref / dt_kv, 1),
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0,
                         "achieved": round(2.0 * n_lin / (step_ms * 1e-3) / 1e9, 1),
                         "frac": round(2.0 * n_lin / (step_ms * 1e-3) / 8e12, 4),
                         "note": "bf16 Linear weights read once per decode step; host launch overhead included"}}), flush=True)


if __name__ == "__main__":
    main()
