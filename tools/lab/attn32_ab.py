#!/usr/bin/env python3
"""LAB: attention forward, 32-row-tile kernel (attn_fwd32_kernel) against the 16-row kernel, same process, alternating:
microseconds per launch (20 back-to-back launches between two events, median of 9), random operands."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from melspec_gpt_vqvae_amd import _ffi, ops

DEV = "cuda:0"
L = _ffi.lib()


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


def main():
    for B, H, T, nu in ((128, 16, 265, 0), (128, 23, 265, 265), (128, 23, 265, 0)):
        C = 64 * H
        qkv = (0.5 * torch.randn(B * T, 3 * C, device=DEV)).to(torch.bfloat16)
        q, k, v = qkv[:, C:2 * C], qkv[:, :C], qkv[:, 2 * C:]
        for p in (0.5, 0.0):
            row = dict(B=B, H=H, T=T, n_unmasked=nu, dropout=p)
            for rnd in range(2):
                for mode in (1, 0):
                    L.melgpt_set_attn_fwd32(mode)
                    row[f"us_{'fwd32' if mode else 'fwd16'}_r{rnd}"] = round(us(lambda: ops.attn_fwd(q, k, v, H, B=B, T=T, n_unmasked=nu, drop_p=p, seed=1, stream_id=0)), 1)
            L.melgpt_set_attn_fwd32(1)
            bytes_alg = 4 * B * T * C * 2
            row["hbm_floor_us_at_8TBs"] = round(bytes_alg / 8e12 * 1e6, 1)
            row["frac_hbm_fwd32"] = round(bytes_alg / (row["us_fwd32_r1"] * 1e-6) / 8e12, 3)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
