#!/usr/bin/env python3
"""Decode-step linears at 16 .. 128 rows: the register-pipelined melgpt_linear_skinny against the LDS-resident
melgpt_linear_lds (ops.LDS_LINEAR_MIN_ROWS switches), in-stream microseconds per launch (200 back-to-back launches
between two events), the four layer shapes of the VAS block + the head.  One JSON line per (rows, shape)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from melspec_gpt_vqvae_amd import ops

DEV = "cuda:0"
SHAPES = [("qkv+ln", 3072, 1024, True, 0, False), ("proj+res", 1024, 1024, False, 0, True),
          ("fc1+ln+gelu", 4096, 1024, True, 1, False), ("fc2+res", 1024, 4096, False, 0, True), ("head+ln", 128, 1024, True, 0, False)]


def timed(fn, n=200):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for M in (16, 32, 64, 128):
    for name, N, K, ln, act, res in SHAPES:
        x = torch.randn(M, K, device=DEV).bfloat16()
        w = (0.05 * torch.randn(N, K, device=DEV)).bfloat16()
        b = torch.randn(N, device=DEV)
        r = torch.randn(M, N, device=DEV).bfloat16() if res else None
        lnp = (torch.ones(K, device=DEV), torch.zeros(K, device=DEV), 1e-5) if ln else None
        out = {}
        for tag, thr in (("skinny", 999), ("lds", 1)):
            ops.LDS_LINEAR_MIN_ROWS = thr
            out[tag] = round(timed(lambda: ops.linear_rows(x, w, bias=b, act=act, residual=r, ln=lnp)), 2)
        ops.LDS_LINEAR_MIN_ROWS = 999
        ya = ops.linear_rows(x, w, bias=b, act=act, residual=r, ln=lnp).float()
        ops.LDS_LINEAR_MIN_ROWS = 1
        yb = ops.linear_rows(x, w, bias=b, act=act, residual=r, ln=lnp).float()
        err = float((ya - yb).abs().max() / ya.abs().max())
        print(json.dumps({"rows": M, "layer": name, "N": N, "K": K, "us_skinny": out["skinny"], "us_lds": out["lds"],
                          "weight_MB": round(N * K * 2 / 1e6, 2), "rel_diff": round(err, 5)}), flush=True)
