#!/usr/bin/env python3
"""LAB: the step's streaming kernels at the training size (ms per launch, TB/s of algorithmic bytes): fused AdamW over the
class-GPT's 302.85 M parameters (30 bytes per parameter), LayerNorm forward / backward on 33 920 x 1024 bf16 rows."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from melspec_gpt_vqvae_amd import ops
from tools.lab.epi_ab import ms

DEV = "cuda:0"


def main():
    out = {"ln_bwd_waves": os.environ.get("MELGPT_LN_BWD_WAVES", "default")}
    n = 302_854_144
    p = torch.randn(n, device=DEV); g = torch.randn(n, device=DEV) * 1e-3
    m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    pb = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    t = ms(lambda: ops.adamw(p, g, m, v, lr=1e-6, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01, step=3, param_bf16=pb), reps=5)
    out["adamw 302.85 M"] = [round(t, 4), round(30.0 * n / t / 1e9, 2)]
    del p, g, m, v, pb
    for M, C in ((33920, 1024), (33920, 1472)):
        ln_rows(out, M, C)
    print(json.dumps(out), flush=True)


def ln_rows(out, M, C):
    x = torch.randn(M, C, device=DEV).to(torch.bfloat16); dy = torch.randn(M, C, device=DEV).to(torch.bfloat16)
    add = torch.randn(M, C, device=DEV).to(torch.bfloat16)
    gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    t = ms(lambda: ops.layernorm_fwd(x, gamma, beta))
    out[f"layernorm fwd {M}x{C}"] = [round(t, 4), round(2 * M * C * 2 / t / 1e9, 2)]
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta)
    t = ms(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, add_in=add, dgamma=dg, dbeta=db))
    out[f"layernorm bwd + add + dgamma {M}x{C}"] = [round(t, 4), round(4 * M * C * 2 / t / 1e9, 2)]
    t = ms(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, add_in=add, dgamma=dg, dbeta=db, mask=(0.5, 3, 1)))
    out[f"layernorm bwd + add + dgamma + masked copy {M}x{C}"] = [round(t, 4), round(5 * M * C * 2 / t / 1e9, 2)]


if __name__ == "__main__":
    main()
