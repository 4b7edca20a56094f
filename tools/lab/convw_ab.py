#!/usr/bin/env python3
"""Lab: the persistent fused GroupNorm+swish+conv3x3 kernel on random data (64 x 80 x 848 x 128, bf16), the four flavours
a ResnetBlock launches (with / without residual, with / without output statistics), 20 launches each, HIP events.
A/B of two library builds: MELGPT_LAB_LIB=... python tools/lab/convw_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from melspec_gpt_vqvae_amd import ops

torch.manual_seed(0)
B, H, W, C = 64, 80, 848, 128
x = torch.randn(B, H, W, C, device="cuda").bfloat16()
res = torch.randn(B, H, W, C, device="cuda").bfloat16()
w = (torch.randn(C, 3, 3, C, device="cuda") * 0.03).bfloat16()
bias = torch.randn(C, device="cuda")
gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
stats = ops.groupnorm_stats(x, 1e-6)


def t(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


fl = 2.0 * B * H * W * C * 9 * C
out = []
for name, fn in [("plain", lambda: ops.conv3x3_gn(x, stats, gamma, beta, w, bias)),
                 ("+res", lambda: ops.conv3x3_gn(x, stats, gamma, beta, w, bias, residual=res)),
                 ("+stats", lambda: ops.conv3x3_gn_with_out_stats(x, stats, gamma, beta, w, bias, 1e-6)),
                 ("+res+stats", lambda: ops.conv3x3_gn_with_out_stats(x, stats, gamma, beta, w, bias, 1e-6, residual=res))]:
    ms = t(fn)
    out.append(f"{name} {ms:.3f} ms ({fl / ms / 1e9:.0f} TFLOP/s)")
print(" | ".join(out), "|", os.path.basename(os.environ.get("MELGPT_LAB_LIB", "in-tree")))
