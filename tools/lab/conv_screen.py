#!/usr/bin/env python3
"""LAB: race screen of the fused GroupNorm + swish + conv3x3 kernels (wave-specialised: LDS counters, hand-counted waits) -
the same launch SCREEN times, output (and the output statistics of the stats variant) compared with the first launch's."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from melspec_gpt_vqvae_amd import ops

DEV = "cuda:0"
N_SCREEN = int(os.environ.get("SCREEN", "1500"))


def screen(name, fn):
    first = [t.clone() for t in fn()]
    bad = torch.zeros((), dtype=torch.int32, device=DEV)
    for _ in range(N_SCREEN):
        for a, b in zip(fn(), first):
            bad += (a != b).any().to(torch.int32)
    print(f"{name:64s} differing launches {int(bad)} of {N_SCREEN}", flush=True)


def main():
    g = torch.Generator(device=DEV).manual_seed(3)
    for B, H, W in ((6, 80, 848), (12, 40, 424), (3, 80, 848)):
        C = 128
        x = (torch.randn(B, H, W, C, device=DEV, generator=g)).to(torch.bfloat16)
        res = (torch.randn(B, H, W, C, device=DEV, generator=g)).to(torch.bfloat16)
        w = (torch.randn(C, 3, 3, C, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
        bias = torch.randn(C, device=DEV, generator=g) * 0.1
        gamma = torch.rand(C, device=DEV, generator=g) + 0.5
        beta = torch.randn(C, device=DEV, generator=g) * 0.1
        stats = ops.groupnorm_stats(x, 1e-6)
        screen(f"conv3x3+gn {B}x{H}x{W} 128->128", lambda: [ops.conv3x3_gn(x, stats, gamma, beta, w, bias, swish=True)])
        screen(f"conv3x3+gn + residual + output stats {B}x{H}x{W}",
               lambda: (lambda r: [r[0], r[1][0], r[1][1]])(ops.conv3x3_gn_with_out_stats(x, stats, gamma, beta, w, bias, 1e-6, swish=True, residual=res)))


if __name__ == "__main__":
    main()
