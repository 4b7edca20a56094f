#!/usr/bin/env python3
"""Micro-benchmark of the MFMA GEMM / conv kernel on the shapes of the hot path (development aid).
Random (gaussian) operands, interleaved rounds, median of per-launch HIP-event times."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from melspec_gpt_vqvae_amd import ops

DEV = "cuda:0"


def time_fn(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
    M = 33920
    shapes = [("fwd qkv", "nt", M, 3072, 1024), ("fwd proj", "nt", M, 1024, 1024), ("fwd fc1", "nt", M, 4096, 1024),
              ("fwd fc2", "nt", M, 1024, 4096), ("fwd head", "nt", M, 128, 1024),
              ("dgrad fc2", "nn", M, 4096, 1024), ("dgrad fc1", "nn", M, 1024, 4096), ("dgrad qkv", "nn", M, 1024, 3072),
              ("wgrad fc1", "tn", 4096, 1024, M), ("wgrad fc2", "tn", 1024, 4096, M), ("wgrad qkv", "tn", 3072, 1024, M),
              ("wgrad proj", "tn", 1024, 1024, M), ("square 4096", "nt", 4096, 4096, 4096)]
    tot_f = tot_t = 0.0
    for name, form, m, n, k in shapes:
        if form == "nt":
            a = torch.randn(m, k, device=DEV).to(dt)
            b = torch.randn(n, k, device=DEV).to(dt)
            fn = lambda: ops.gemm(a, b)
        elif form == "nn":
            a = torch.randn(m, k, device=DEV).to(dt)
            b = torch.randn(k, n, device=DEV).to(dt)
            fn = lambda: ops.gemm(a, b, b_kmajor=True)
        else:
            a = torch.randn(k, m, device=DEV).to(dt)
            b = torch.randn(k, n, device=DEV).to(dt)
            out = torch.empty(m, n, device=DEV)
            fn = lambda: ops.gemm(a, b, a_kmajor=True, b_kmajor=True, out=out)
        ms = time_fn(fn)
        fl = 2.0 * m * n * k
        tot_f += fl
        tot_t += ms
        print(f"{name:14s} {form} M={m:6d} N={n:5d} K={k:6d}  {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s", flush=True)
    print(f"{'weighted':14s} {tot_f / tot_t / 1e9:8.1f} TFLOP/s")
    # conv: level-0 3x3 128->128 at B=16
    B, H, W, C = 16, 80, 848, 128
    x = torch.randn(B, H, W, C, device=DEV).to(dt)
    w = torch.randn(C, 3, 3, C, device=DEV).to(dt)
    ms = time_fn(lambda: ops.conv2d_nhwc(x, w, None))
    fl = 2.0 * B * H * W * C * 9 * C
    print(f"conv3x3 128->128 80x848 B={B}: {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s")


if __name__ == "__main__":
    main()
