#!/usr/bin/env python3
"""(lab library: MELGPT_LAB_LIB=tools/lab/bin/libmelgpt_r04gemm.so, built by tools/lab/build_lab_lib.py from tools/lab/gemm256_r04.hip)
A/B of the persistent GEMM's 32-deep K units / ten-slot ring (MELGPT_GEMM_KU32=1) against the 64-deep / five-slot
ring (=0) on the step's four weight-gradient shapes: RANDOM operands, both arms interleaved in ONE process, the step's
own entry point (ops.wgrad: split-K batches + bias row sums), median and minimum of per-launch HIP-event times.
Also checks that both arms give the same weight gradient (to f32 summation order) and bias gradient."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from melspec_gpt_vqvae_amd import ops

DEV = "cuda:0"
M = 33920
SHAPES = [("fc1", 4096, 1024), ("fc2", 1024, 4096), ("qkv", 3072, 1024), ("proj", 1024, 1024)]
ROUNDS = int(os.environ.get("ROUNDS", "7"))
REPS = int(os.environ.get("REPS", "6"))


def main():
    torch.manual_seed(0)
    rows = []
    for name, N, K in SHAPES:
        dy = torch.randn(M, N, device=DEV).to(torch.bfloat16)
        x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        out = {a: torch.empty(N, K, device=DEV) for a in "01"}
        bias = {a: torch.empty(N, device=DEV) for a in "01"}
        times = {"0": [], "1": []}
        for arm in "01":                                   # warm-up + results
            os.environ["MELGPT_GEMM_KU32"] = arm
            ops.wgrad(dy, x, out[arm], False, bias_out=bias[arm], bias_accumulate=False)
        torch.cuda.synchronize()
        ref = (dy.float().T @ x.float())
        e0 = float((out["0"] - ref).abs().max() / ref.abs().max())
        e1 = float((out["1"] - ref).abs().max() / ref.abs().max())
        eb = float((bias["1"] - bias["0"]).abs().max() / bias["0"].abs().max())
        for r in range(ROUNDS):
            for arm in ("01" if r % 2 == 0 else "10"):
                os.environ["MELGPT_GEMM_KU32"] = arm
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(REPS):
                    ops.wgrad(dy, x, out[arm], False, bias_out=bias[arm], bias_accumulate=False)
                e.record()
                torch.cuda.synchronize()
                times[arm].append(s.elapsed_time(e) / REPS)
        fl = 2.0 * M * N * K
        rec = {"shape": f"wgrad {name} TN {N}x{K}x{M}", "err_ku64": e0, "err_ku32": e1, "bias_diff": eb}
        for arm, tag in (("0", "ku64"), ("1", "ku32")):
            t = sorted(times[arm])
            rec[tag + "_ms_med"] = round(t[len(t) // 2], 4)
            rec[tag + "_ms_min"] = round(t[0], 4)
            rec[tag + "_tflops_med"] = round(fl / t[len(t) // 2] / 1e9, 1)
        rec["speedup_med"] = round(rec["ku64_ms_med"] / rec["ku32_ms_med"], 4)
        rows.append(rec)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
