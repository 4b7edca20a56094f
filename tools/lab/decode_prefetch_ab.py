#!/usr/bin/env python3
"""LAB: KV-cached greedy sampling of 265 tokens (class-GPT VAS, 16-bit lane, one replayed graph per token) with / without the
warm-up workgroups of the weight-streaming Linear nodes (minGPT.DECODE_PREFETCH; csrc/decode.hip gemv_rows_kernel): ms per
run, median of 5, and whether the tokens equal the first arm's (they must: the extra workgroups only read).
usage: ARMS=0,1,0,1 decode_prefetch_ab.py <batch>[,<batch>...]     (every arm in ONE process, a fresh graph per run)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch

import synth
import melspec_gpt_vqvae_amd.transformer.minGPT as mg
from melspec_gpt_vqvae_amd.transformer.minGPT import Lit_minGPT, set_compute_dtype

DEV = "cuda:0"


def main():
    batches = [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "1,64").split(",")]
    arms = [int(a) for a in os.environ.get("ARMS", "0,1,0,1").split(",")]
    args = synth.gpt_args(n_layer=24, n_head=16, n_embd=1024, reconstruct_spec="", device=DEV, batch_size=2, learning_rate=1e-6)
    torch.manual_seed(1)
    lit = Lit_minGPT(args).to(DEV).eval()
    set_compute_dtype(lit.transformer, torch.bfloat16)
    for B in batches:
        c = torch.randint(0, 8, (B, 1), device=DEV)
        x0 = torch.zeros(B, 0, dtype=torch.int64, device=DEV)
        ref = None
        for wgs in arms:
            mg.DECODE_PREFETCH = bool(wgs)
            lit.sample(x0, c, steps=16, sample=False)
            torch.cuda.synchronize()
            ts, xs = [], None
            for _ in range(5):
                t0 = time.perf_counter()
                xs, _ = lit.sample(x0, c, steps=265, sample=False)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            ts.sort()
            if ref is None:
                ref = xs.clone()
            print(json.dumps({"batch": B, "warm_up_workgroups": bool(wgs), "ms_265_tokens": round(1e3 * ts[2], 2),
                              "ms_per_token": round(1e3 * ts[2] / 265, 4), "min_ms": round(1e3 * ts[0], 2),
                              "same_tokens_as_first_arm": bool(torch.equal(xs, ref))}), flush=True)


if __name__ == "__main__":
    main()
