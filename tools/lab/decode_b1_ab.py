#!/usr/bin/env python3
"""LAB: sampling 265 tokens at 1 / 2 / 4 sequences (class-GPT VAS, 16-bit lane, one replayed graph per token), ms per run
(median of 5).  A/B knob: MELGPT_DECODE_QKV_ATTN=0 (qkv projection and attention step as two launches)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import synth
from melspec_gpt_vqvae_amd import _ffi
from melspec_gpt_vqvae_amd.transformer.minGPT import Lit_minGPT, set_compute_dtype
DEV = "cuda:0"
args = synth.gpt_args(n_layer=24, n_head=16, n_embd=1024, reconstruct_spec="", device=DEV, batch_size=2, learning_rate=1e-6)
lit = Lit_minGPT(args).to(DEV).eval()
set_compute_dtype(lit.transformer, _ffi.HALF_DTYPE)
out = {"qkv_attn_fused": os.environ.get("MELGPT_DECODE_QKV_ATTN", "1")}
for B in (1, 2, 4):
    c = torch.randint(0, 8, (B, 1), device=DEV)
    x0 = torch.zeros(B, 0, dtype=torch.int64, device=DEV)
    lit.sample(x0, c, steps=8, sample=False)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        xs, _ = lit.sample(x0, c, steps=265, sample=False)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    out[f"B={B}"] = {"ms_265_tokens": round(ts[2], 2), "ms_per_token": round(ts[2] / 265, 4), "checksum": int(xs.sum())}
print(json.dumps(out), flush=True)
