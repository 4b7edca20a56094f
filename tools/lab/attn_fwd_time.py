#!/usr/bin/env python3
"""LAB: the 16-row attention forward at the training shape, us per launch (20 back-to-back launches, median of 9)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from melspec_gpt_vqvae_amd import _ffi, ops


def us(fn, reps=20, iters=9):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / reps * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


DEV = "cuda:0"
B, H, T = 128, 16, 265
C = 64 * H
qkv = (0.5 * torch.randn(B * T, 3 * C, device=DEV)).to(torch.bfloat16)
q, k, v = qkv[:, C:2 * C], qkv[:, :C], qkv[:, 2 * C:]
_ffi.lib().melgpt_set_attn_fwd32(0)
print(json.dumps({"lib": os.path.basename(os.environ.get("MELGPT_LAB_LIB", "production")),
                  "fwd16_causal_p0.5_us": round(us(lambda: ops.attn_fwd(q, k, v, H, B=B, T=T, n_unmasked=0, drop_p=0.5, seed=1, stream_id=0)), 1),
                  "fwd16_causal_p0_us": round(us(lambda: ops.attn_fwd(q, k, v, H, B=B, T=T, n_unmasked=0, drop_p=0.0, seed=1, stream_id=0)), 1)}))
