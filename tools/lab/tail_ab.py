#!/usr/bin/env python3
"""LAB: what the last, partial round of the persistent GEMM costs on the N = 1024 shapes of a Block (ms per launch, 20
back-to-back launches between two events, median of 7, random operands): M = 33 920 (2.07 rounds of 256-row tiles on 256
CUs) against M = 32 768 (exactly two rounds) scaled by the rows.  Run once per tile height (MELGPT_GEMM_TM=6|8, unset)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from melspec_gpt_vqvae_amd import ops
from tools.lab.epi_ab import ms

DEV = "cuda:0"


def main():
    g = torch.Generator(device=DEV).manual_seed(7)
    rnd = lambda *s, sc=0.5: (torch.randn(*s, device=DEV, generator=g) * sc).to(torch.bfloat16)
    out = {"tm": os.environ.get("MELGPT_GEMM_TM", "auto"), "tail": os.environ.get("MELGPT_GEMM_TAIL", "1"),
           "split": os.environ.get("MELGPT_GEMM_TAIL_SPLIT", "default")}
    for N, K in ((1024, 4096), (1024, 3072), (1024, 1024), (3072, 1024), (4096, 1024)):
        for M in (33920, 32768):
            x = rnd(M, K)
            w_nt, w_nn = rnd(N, K, sc=0.25), rnd(K, N, sc=0.25)
            fl = 2.0 * M * N * K
            t1 = ms(lambda: ops.gemm(x, w_nt))
            t2 = ms(lambda: ops.gemm(x, w_nn, b_kmajor=True))
            out[f"NT {M}x{N}x{K}"] = [round(t1, 4), round(fl / t1 / 1e9, 1)]
            out[f"NN {M}x{N}x{K}"] = [round(t2, 4), round(fl / t2 / 1e9, 1)]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
