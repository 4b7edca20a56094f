#!/usr/bin/env python3
"""(lab library: MELGPT_LAB_LIB=tools/lab/bin/libmelgpt_r04gemm.so, built by tools/lab/build_lab_lib.py from tools/lab/gemm256_r04.hip)
A/B of the persistent GEMM's four-wave form (MELGPT_GEMM_W4=1: one wave per SIMD, 128 x 128 accumulator blocks in the
accumulator half of the register file) against the shipped eight-wave form on row-major x row-major shapes with the plain
bf16 epilogue: RANDOM operands, arms interleaved in ONE process, median / minimum of per-launch HIP-event times; outputs
of the two arms compared bit for bit."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from melspec_gpt_vqvae_amd import ops

DEV = "cuda:0"
SHAPES = [("qkv fwd", 33920, 3072, 1024), ("fc1 plain", 33920, 4096, 1024), ("fc2 plain", 33920, 1024, 4096),
          ("square 8192", 8192, 8192, 8192)]
ROUNDS = int(os.environ.get("ROUNDS", "7"))
REPS = int(os.environ.get("REPS", "6"))


def main():
    torch.manual_seed(0)
    os.environ["MELGPT_GEMM_TM"] = "8"      # both arms on 256-row tiles (the four-wave form has no 192-row variant)
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        b = torch.randn(N, K, device=DEV).to(torch.bfloat16)
        bias = torch.randn(N, device=DEV)
        outs, times = {}, {"0": [], "1": []}
        for arm in "01":
            os.environ["MELGPT_GEMM_W4"] = arm
            outs[arm] = ops.gemm(a, b, bias=bias)
        torch.cuda.synchronize()
        same = bool(torch.equal(outs["0"], outs["1"]))
        for r in range(ROUNDS):
            for arm in ("01" if r % 2 == 0 else "10"):
                os.environ["MELGPT_GEMM_W4"] = arm
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(REPS):
                    ops.gemm(a, b, bias=bias)
                e.record()
                torch.cuda.synchronize()
                times[arm].append(s.elapsed_time(e) / REPS)
        fl = 2.0 * M * N * K
        rec = {"shape": f"{name} NT {M}x{N}x{K}", "bit_identical": same}
        for arm, tag in (("0", "w8"), ("1", "w4")):
            t = sorted(times[arm])
            rec[tag + "_ms_med"] = round(t[len(t) // 2], 4)
            rec[tag + "_ms_min"] = round(t[0], 4)
            rec[tag + "_tflops_med"] = round(fl / t[len(t) // 2] / 1e9, 1)
        rec["speedup_med"] = round(rec["w8_ms_med"] / rec["w4_ms_med"], 4)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
