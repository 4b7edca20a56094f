#!/usr/bin/env python3
"""MelGAN generator latency: mel (B, 80, 848) -> waveform (B, 1, 217 088), ngf = 32, 3 residual layers
(the reference's vocoder configuration, callbacks/GPT_callbacks.py:66-79), bf16 and f32 lanes.  One JSON line per
case; FLOPs counted from the convolution shapes."""
import json
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch

from melspec_gpt_vqvae_amd.vocoder import Generator

DEV = "cuda:0"


def flops(T, ngf=32, n_res=3, n_mel=80):
    f, L, mult = 2.0 * T * n_mel * 16 * ngf * 7, T, 16
    for r in (8, 8, 2, 2):
        cin, cout = mult * ngf, mult * ngf // 2
        L *= r
        f += 2.0 * L * cin * cout * 2                      # two taps reach every output sample
        f += n_res * 2.0 * L * cout * cout * (3 + 1 + 1)
        mult //= 2
    return f + 2.0 * L * ngf * 7


def main():
    torch.manual_seed(0)
    g = Generator(80, 32, 3).to(DEV).eval()
    for dt in (torch.bfloat16, torch.float32):
        for m in g.modules():
            object.__setattr__(m, "compute_dtype", dt)
        for B in ((1, 8, 64) if dt == torch.bfloat16 else (1, 8)):
            x = torch.randn(B, 80, 848, device=DEV)
            g(x)
            torch.cuda.synchronize()
            n = 5 if B < 64 else 3
            t0 = time.perf_counter()
            for _ in range(n):
                y = g(x)
            torch.cuda.synchronize()
            dt_s = (time.perf_counter() - t0) / n
            print(json.dumps({"bench": "MelGAN generator 848 frames -> 217088 samples", "dtype": str(dt).split(".")[-1],
                              "batch": B, "ms": round(1e3 * dt_s, 3), "clips_per_s": round(B / dt_s, 1),
                              "x_realtime": round(B * 217088 / 22050 / dt_s, 1),
                              "tflops": round(B * flops(848) / dt_s / 1e12, 2), "out_shape": list(y.shape)}), flush=True)


if __name__ == "__main__":
    main()
