#!/usr/bin/env python3
"""LAB: race screen of the ping-pong GEMM's less common forms - the same launch SCREEN times, every output compared with
the first one: K-major operands off the 128 boundaries (XL widths), single-round grids, grids with reserved CUs, ragged
edges with a half-height last round, the weight gradients' split-K batches."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from melspec_gpt_vqvae_amd import _ffi, ops

DEV = "cuda:0"
N_SCREEN = int(os.environ.get("SCREEN", "3000"))
L = _ffi.lib()


def launches():
    r, p = ctypes.c_longlong(0), ctypes.c_longlong(0)
    L.melgpt_gemm_loop_launches(ctypes.byref(r), ctypes.byref(p))
    return p.value


def screen(name, fn):
    p0 = launches()
    first = [t.clone() for t in fn()]
    bad = torch.zeros((), dtype=torch.int32, device=DEV)
    for _ in range(N_SCREEN):
        for a, b in zip(fn(), first):
            bad += (a != b).any().to(torch.int32)
    print(f"{name:58s} ping-pong launches {launches() - p0:6d}  differing launches {int(bad)}", flush=True)


def main():
    g = torch.Generator(device=DEV).manual_seed(1)
    rnd = lambda *s: (torch.randn(*s, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    M = 33920
    a14, a58 = rnd(M, 1472), rnd(M, 5888)
    w_fc2 = rnd(1472, 5888)      # (out, in): dX = dY W  ->  K-major B with N = in
    w_fc1 = rnd(5888, 1472)
    screen("XL dgrad NN 33920x5888x1472 (N % 128 == 0, K ragged)", lambda: [ops.gemm(a14, w_fc2, b_kmajor=True)])
    screen("XL dgrad NN 33920x1472x5888 (EDGE)", lambda: [ops.gemm(a58, w_fc1, b_kmajor=True)])
    screen("XL fwd NT 33920x1472x5888", lambda: [ops.gemm(a58, w_fc2)])

    def wg(dy, x):
        n, k = dy.shape[1], x.shape[1]
        w, b = torch.empty(n, k, device=DEV), torch.empty(n, device=DEV)
        ops.wgrad(dy, x, w, False, bias_out=b)
        return [w, b]
    screen("XL wgrad 1472x5888 (EDGE A)", lambda: wg(a14, a58))
    screen("XL wgrad 5888x1472 (EDGE B)", lambda: wg(a58, a14))
    a_s, b_s = rnd(9000, 1024), rnd(2048, 1024)
    screen("single round NT 9000x2048x1024 (288 tiles? / fewer than CUs)", lambda: [ops.gemm(a_s, b_s)])
    a_r, b_r = rnd(33920 - 77, 1000), rnd(4096 - 24, 1000)
    for res in (16, 8, 0):
        L.melgpt_set_reserved_cus(res)
        screen(f"ragged NT {a_r.shape[0]}x{b_r.shape[0]}x1000, {res} reserved CUs", lambda: [ops.gemm(a_r, b_r)])
    L.melgpt_set_reserved_cus(0)


if __name__ == "__main__":
    main()
