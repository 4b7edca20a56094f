#!/usr/bin/env python3
"""LAB: per-tile s_memtime stamps of gemm8p's workgroup 17 (lab library built with -DP8_LAB, selected by MELGPT_LAB_LIB).
Prints, per tile: K loop, drain, epilogue, gap to the next tile (cycles of the 100 MHz s_memtime counter x 21 ~ shader cycles
at 2.1 GHz: reported raw) and the per-K-tile times."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from melspec_gpt_vqvae_amd import _ffi, ops

M, N, K = (int(v) for v in os.environ.get("SHAPE", "33920,4096,1024").split(","))
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
KM = os.environ.get("B_KMAJOR", "0") == "1"
b = (torch.randn(*((K, N) if KM else (N, K)), device="cuda") * 0.02).to(torch.bfloat16)
MODE = os.environ.get("MODE", "plain")      # plain | bias | drop_res | gelu_dact | mul : the epilogue the launch carries
bias = torch.randn(N, device="cuda") * 0.1
res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
kw = {"plain": {}, "bias": dict(bias=bias), "drop_res": dict(bias=bias, drop_p=0.5, seed=3, stream_id=2, residual=res),
      "gelu_dact": dict(bias=bias, act=ops.ACT_GELU_DACT, pre_out=pre), "mul": dict(act=ops.ACT_MUL, residual=res)}[MODE]
for _ in range(3):
    ops.gemm(a, b, b_kmajor=KM, **kw)
torch.cuda.synchronize()
print(f"shape {M}x{N}x{K} b_kmajor={int(KM)} mode={MODE}")
L = _ffi.lib()
buf = (ctypes.c_ulonglong * (64 * 24))()
L.melgpt_p8_dbg.argtypes = [ctypes.c_void_p]
assert L.melgpt_p8_dbg(buf) == 0
v = list(buf)
nu = (K + 63) // 64
for ti in range(10):
    r = v[ti * 24:(ti + 1) * 24]
    if r[0] == 0:
        break
    nxt = v[(ti + 1) * 24]
    kt = [r[4 + u] - (r[4 + u - 1] if u else r[0]) for u in range(min(nu, 20))]
    print(f"tile {ti}: kloop {r[1] - r[0]} drain {r[2] - r[1]} epilogue {r[3] - r[2]} gap {nxt - r[3] if nxt else -1} | per K tile {kt}")
print("last plan() took", v[64 * 24 - 1])
