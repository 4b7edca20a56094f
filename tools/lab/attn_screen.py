#!/usr/bin/env python3
"""LAB: repeat screen of the attention kernels (forward, single-pass backward): the same launch SCREEN times, outputs
compared with the first launch's."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from melspec_gpt_vqvae_amd import ops

DEV = "cuda:0"
N_SCREEN = int(os.environ.get("SCREEN", "2000"))
B, H, T, C = 32, 16, 265, 1024
g = torch.Generator(device=DEV).manual_seed(5)
q, k, v, do = ((torch.randn(B * T, C, device=DEV, generator=g) * 0.5).to(torch.bfloat16) for _ in range(4))
for p_drop in (0.5, 0.0):
    o, lse, _ = ops.attn_fwd(q, k, v, H, B=B, T=T, drop_p=p_drop, seed=1, stream_id=0)
    first_f = [o.clone(), lse.clone()]
    first_b = [t.clone() for t in ops.attn_bwd(q, k, v, o, do, lse, H, B=B, T=T, drop_p=p_drop, seed=1, stream_id=0)]
    bad = torch.zeros(2, dtype=torch.int32, device=DEV)
    for _ in range(N_SCREEN):
        o2, lse2, _ = ops.attn_fwd(q, k, v, H, B=B, T=T, drop_p=p_drop, seed=1, stream_id=0)
        bad[0] += ((o2 != first_f[0]).any() | (lse2 != first_f[1]).any()).to(torch.int32)
        for a, b in zip(ops.attn_bwd(q, k, v, o, do, lse, H, B=B, T=T, drop_p=p_drop, seed=1, stream_id=0), first_b):
            bad[1] += (a != b).any().to(torch.int32)
    print(f"attention B {B} H {H} T {T} dropout {p_drop}: differing forward launches {int(bad[0])}, backward tensors {int(bad[1])} of {N_SCREEN}", flush=True)
