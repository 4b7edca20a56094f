#!/usr/bin/env python3
"""LAB: does a warm cache make a decode node faster?  The four Linear nodes of a VAS block at 1 / 64 rows, each captured
as a chain of 48 dependent launches in ONE HIP graph and replayed (as the sampling graph runs them):
  cold: the 48 launches walk 48 DISTINCT weight tensors (the layer's bytes x 48: every launch first-touches HBM; with
        fc1 / fc2 that is 400 MB between two uses of a tensor - beyond the 256 MB memory-side cache),
  warm: all 48 launches use the SAME weight tensor (after the first: L2 / memory-side cache hits).
us per node = replay time / 48, median of 9 replays.  The difference bounds what ANY weight prefetch can return."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from melspec_gpt_vqvae_amd import ops

DEV = "cuda:0"
SHAPES = [("qkv+ln", 3072, 1024, True, 0, False), ("proj+res", 1024, 1024, False, 0, True),
          ("fc1+ln+gelu", 4096, 1024, True, 1, False), ("fc2+res", 1024, 4096, False, 0, True)]
NCH = 48


def replay_us(g, n=9):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1) / NCH)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    for M in (1, 64):
        for name, N, K, ln, act, res in SHAPES:
            ws = [(0.05 * torch.randn(N, K, device=DEV)).bfloat16() for _ in range(NCH)]
            b = torch.randn(N, device=DEV)
            lnp = (torch.ones(K, device=DEV), torch.zeros(K, device=DEV), 1e-5) if ln else None
            x0 = torch.randn(M, K, device=DEV).bfloat16()
            r = torch.randn(M, N, device=DEV).bfloat16() if res else None
            out = {}
            for tag in ("cold", "warm"):
                def chain():
                    # (dependent launches: each node's x is the previous node's output where the shapes allow it, else x0 -
                    # stream order makes them dependent either way, as the sampling graph's nodes are)
                    y = None
                    for i in range(NCH):
                        w = ws[i] if tag == "cold" else ws[0]
                        y = ops.linear_rows(x0, w, bias=b, act=act, residual=r, ln=lnp)
                    return y
                chain()
                torch.cuda.synchronize()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    chain()
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    chain()
                g.replay()
                torch.cuda.synchronize()
                out[tag] = round(replay_us(g), 2)
                del g
            print(json.dumps({"rows": M, "node": name, "weight_MB": round(N * K * 2 / 1e6, 2), "us_cold": out["cold"],
                              "us_warm": out["warm"], "warm_saves_us": round(out["cold"] - out["warm"], 2)}), flush=True)
            del ws


if __name__ == "__main__":
    main()
