#!/usr/bin/env python3
"""LAB: the four epilogue-carrying GEMM shapes of a Block (+ their plain twins) at the training size, ms per launch
(20 back-to-back launches between two events, median of 7), random operands; the R operands also COLD (pools larger than the
memory-side cache).  A/B: the library named by MELGPT_LAB_LIB against the in-tree one."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from melspec_gpt_vqvae_amd import ops

DEV = "cuda:0"


def ms(fn, reps=20, iters=7):
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
        ts.append(s.elapsed_time(e) / reps)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    g = torch.Generator(device=DEV).manual_seed(7)
    rnd = lambda *s, sc=0.5: (torch.randn(*s, device=DEV, generator=g) * sc).to(torch.bfloat16)
    M, C, F = 33920, 1024, 4096
    x, x4 = rnd(M, C), rnd(M, F)
    w1, w2, wp = rnd(F, C, sc=0.25), rnd(C, F, sc=0.25), rnd(C, C, sc=0.25)
    b4, b1 = torch.randn(F, device=DEV) * 0.1, torch.randn(C, device=DEV) * 0.1
    res1, dact = rnd(M, C, sc=1.0), rnd(M, F, sc=1.0)
    # R operands that are COLD, as in the step (written tens of milliseconds before they are read): pools larger than the
    # 256 MB memory-side cache, taken round robin
    res_pool = [rnd(M, C, sc=1.0) for _ in range(8)]
    dact_pool = [rnd(M, F, sc=1.0) for _ in range(3)]
    cnt = [0]

    def nxt(pool):
        cnt[0] += 1
        return pool[cnt[0] % len(pool)]
    pre = torch.empty(M, F, dtype=torch.bfloat16, device=DEV)
    rows = [("fc1 + GELU + derivative", lambda: ops.gemm(x, w1, bias=b4, act=ops.ACT_GELU_DACT, pre_out=pre), 2.0 * M * F * C),
            ("fc1 plain", lambda: ops.gemm(x, w1, bias=b4), 2.0 * M * F * C),
            ("fc2 + dropout + residual", lambda: ops.gemm(x4, w2, bias=b1, drop_p=0.5, seed=3, stream_id=2, residual=res1), 2.0 * M * F * C),
            ("proj + dropout + residual", lambda: ops.gemm(x, wp, bias=b1, drop_p=0.5, seed=3, stream_id=1, residual=res1), 2.0 * M * C * C),
            ("proj plain", lambda: ops.gemm(x, wp, bias=b1), 2.0 * M * C * C),
            ("GELU' dgrad (x saved derivative)", lambda: ops.gemm(x, w2, b_kmajor=True, act=ops.ACT_MUL, residual=dact), 2.0 * M * F * C),
            ("GELU' dgrad, COLD R", lambda: ops.gemm(x, w2, b_kmajor=True, act=ops.ACT_MUL, residual=nxt(dact_pool)), 2.0 * M * F * C),
            ("fc2 + dropout + residual, COLD R", lambda: ops.gemm(x4, w2, bias=b1, drop_p=0.5, seed=3, stream_id=2, residual=nxt(res_pool)), 2.0 * M * F * C),
            ("proj + dropout + residual, COLD R", lambda: ops.gemm(x, wp, bias=b1, drop_p=0.5, seed=3, stream_id=1, residual=nxt(res_pool)), 2.0 * M * C * C),
            ("dgrad plain", lambda: ops.gemm(x, w2, b_kmajor=True), 2.0 * M * F * C),
            # the same epilogues with the R operand cache-resident (ONE row, row stride 0): what is left is not HBM reads
            ("GELU' dgrad, R = one row (cache-resident)", lambda: ops.gemm(x, w2, b_kmajor=True, act=ops.ACT_MUL, residual=dact[:1].expand(M, F)), 2.0 * M * F * C),
            ("proj + dropout + residual, R = one row", lambda: ops.gemm(x, wp, bias=b1, drop_p=0.5, seed=3, stream_id=1, residual=res1[:1].expand(M, C)), 2.0 * M * C * C),
            ("proj + residual (no dropout)", lambda: ops.gemm(x, wp, bias=b1, residual=res1), 2.0 * M * C * C)]
    out = {"lib": os.path.basename(os.environ.get("MELGPT_LAB_LIB", "production"))}
    for name, fn, fl in rows:
        t = ms(fn)
        out[name] = [round(t, 4), round(fl / t / 1e9, 1)]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
