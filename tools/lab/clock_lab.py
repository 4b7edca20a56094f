#!/usr/bin/env python3
"""LAB (diagnostic build): the IN-KERNEL shader clock of the step's big kernels - d(s_memtime) / d(s_memrealtime) x 100 MHz
per workgroup, median over workgroups, read after >= 2 s of back-to-back launches of the shape on random data
(MI355X_MICROARCH "DVFS give-back" item 6).  Run with MELGPT_LAB_LIB=tools/lab/bin/libmelgpt_clock.so
(tools/lab/build_clock_lib.py).  One JSON line per kernel shape; beside the clock: microseconds per launch in this
(stamped) build and the rate that implies."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from melspec_gpt_vqvae_amd import _ffi, ops

DEV = "cuda:0"
SECONDS = float(os.environ.get("CLOCK_SECONDS", "2.0"))
L = ctypes.CDLL(_ffi.LIB_PATH)
assert hasattr(L, "melgpt_clk_gemm8p"), "load the clock-stamp build: MELGPT_LAB_LIB=tools/lab/bin/libmelgpt_clock.so"


def read(sym):
    buf = (ctypes.c_ulonglong * 4096)()
    assert getattr(L, sym)(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 2).astype(np.float64)
    a = a[(a[:, 0] > 0) & (a[:, 1] > 0)]
    return a


ONLY = os.environ.get("CLOCK_ONLY", "")


def clock(name, sym, fn, flops=None):
    if ONLY and ONLY not in name:
        return
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < SECONDS:      # keep the queue full: >= 2 s of back-to-back launches
        for _ in range(50):
            fn()
        n += 50
        if n % 500 == 0:
            torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 50 * 1e3
    a = read(sym)                                  # stamps of the LAST launch
    if len(a) == 0:                                # (the launch took a kernel that carries no stamps in this build)
        print(json.dumps(dict(kernel=name, note="no stamps", us_per_launch=round(us, 1))), flush=True)
        return
    ghz = a[:, 0] / a[:, 1] * 0.1
    row = dict(kernel=name, workgroups=int(len(a)), resident_workgroups_mean=round(float(a[:, 1].sum()) * 0.01 / us, 1),
               in_kernel_clock_GHz=round(float(np.median(ghz)), 3),
               clock_p10_p90=[round(float(np.percentile(ghz, 10)), 3), round(float(np.percentile(ghz, 90)), 3)],
               workgroup_life_us_median=round(float(np.median(a[:, 1])) * 0.01, 1), us_per_launch=round(us, 1))
    if flops:
        row["TFLOPs"] = round(flops / us / 1e6, 1)
        row["frac_of_peak_at_that_clock"] = round(flops / us / 1e6 / (2500.0 * row["in_kernel_clock_GHz"] / 2.4), 3)
    print(json.dumps(row), flush=True)


def main():
    g = torch.Generator(device=DEV).manual_seed(7)
    rnd = lambda *s, sc=0.5: (torch.randn(*s, device=DEV, generator=g) * sc).to(torch.bfloat16)
    M, C, F = 33920, 1024, 4096
    x, x4 = rnd(M, C), rnd(M, F)
    w1, w2, wq = rnd(F, C, sc=0.25), rnd(C, F, sc=0.25), rnd(3 * C, C, sc=0.25)
    bias4, bias1 = torch.randn(F, device=DEV) * 0.1, torch.randn(C, device=DEV) * 0.1
    res1 = rnd(M, C, sc=1.0)
    dact = rnd(M, F, sc=1.0)
    pre = torch.empty(M, F, dtype=torch.bfloat16, device=DEV)
    clock("gemm NT 33920x4096x1024 gelu+dact (fc1)", "melgpt_clk_gemm8p",
          lambda: ops.gemm(x, w1, bias=bias4, act=ops.ACT_GELU_DACT, pre_out=pre), 2.0 * M * F * C)
    clock("gemm NT 33920x1024x4096 drop res (fc2)", "melgpt_clk_gemm8p",
          lambda: ops.gemm(x4, w2, bias=bias1, drop_p=0.5, seed=3, stream_id=2, residual=res1), 2.0 * M * F * C)
    clock("gemm NT 33920x3072x1024 (qkv)", "melgpt_clk_gemm8p", lambda: ops.gemm(x, wq, bias=torch.zeros(3 * C, device=DEV)), 2.0 * M * 3 * C * C)
    clock("gemm NN 33920x4096x1024 mul (GELU' dgrad)", "melgpt_clk_gemm8p",
          lambda: ops.gemm(x, w2, b_kmajor=True, act=ops.ACT_MUL, residual=dact), 2.0 * M * F * C)
    clock("gemm NN 33920x1024x4096 (fc1 dgrad)", "melgpt_clk_gemm8p", lambda: ops.gemm(x4, w1, b_kmajor=True), 2.0 * M * F * C)
    wg, bg = torch.empty(F, C, device=DEV), torch.empty(F, device=DEV)
    clock("gemm TN 4096x1024x8480 b4 +rowsum (fc1 wgrad)", "melgpt_clk_gemm8p", lambda: ops.wgrad(x4, x, wg, False, bias_out=bg), 2.0 * M * F * C)
    a8, b8 = rnd(8192, 8192), rnd(8192, 8192, sc=0.25)
    clock("gemm NT 8192^3", "melgpt_clk_gemm8p", lambda: ops.gemm(a8, b8), 2.0 * 8192 ** 3)
    # fused GroupNorm + swish + conv3x3, 64 tiles of 80 x 848, 128 -> 128
    B, H, W = 64, 80, 848
    xi = rnd(B, H, W, 128, sc=1.0)
    wc = rnd(128, 3, 3, 128, sc=0.05)
    bc, gm, bt = torch.randn(128, device=DEV) * 0.1, torch.rand(128, device=DEV) + 0.5, torch.randn(128, device=DEV) * 0.1
    st = ops.groupnorm_stats(xi, 1e-6)
    clock("conv3x3+gn 64x80x848 128->128", "melgpt_clk_conv_ws", lambda: ops.conv3x3_gn(xi, st, gm, bt, wc, bc, swish=True),
          2.0 * B * H * W * 128 * 9 * 128)
    del xi
    # attention at the training shape
    Bq, Hh, T = 128, 16, 265
    qkv = rnd(Bq * T, 3 * C)
    q, k, v = qkv[:, C:2 * C], qkv[:, :C], qkv[:, 2 * C:]
    full = 4.0 * T * T * C * Bq
    clock("attention forward 128x16x265 dropout 0.5", "melgpt_clk_attn_fwd",
          lambda: ops.attn_fwd(q, k, v, Hh, B=Bq, T=T, drop_p=0.5, seed=1, stream_id=0), full)
    print(json.dumps(dict(kernel="attn_fwd32_kernel", runtime_occupancy_workgroups_per_cu=int(L.melgpt_clk_attn_fwd32_occupancy(T)))), flush=True)
    # phase stamps of one workgroup of attn_fwd32_kernel (head 5 of batch 3), cycles from the workgroup's entry
    if ONLY and ONLY not in "attention forward 128x16x265 dropout 0.5":
        return
    buf = (ctypes.c_ulonglong * 4096)()
    assert L.melgpt_clk_attn32_ph(buf) == 0
    ph = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)[:256].reshape(4, 64)
    t0 = int(ph[:, 0].min())
    for w in range(4):
        n = int(ph[w, 63])
        row = dict(kernel="attn_fwd32 phases", wave=w, staged=int(ph[w, 1] - t0), tiles=[])
        i = 2
        while i + 6 < n:
            st, b_done, p_done, s_done, m_done, pv_done, nkt = (int(x) for x in ph[w, i:i + 7])
            row["tiles"].append(dict(key_tiles=nkt, start=st - t0, bounds=b_done - st, first_pair=p_done - b_done, S_rest=s_done - p_done,
                                     rowmax=m_done - s_done, softmax_pv=pv_done - m_done))
            i += 7
        row["done"] = int(ph[w, i] - t0) if i < n else None
        print(json.dumps(row), flush=True)
    o, lse, _ = ops.attn_fwd(q, k, v, Hh, B=Bq, T=T, drop_p=0.5, seed=1, stream_id=0)
    do = rnd(Bq * T, C)
    clock("attention backward 128x16x265 dropout 0.5 (single pass)", "melgpt_clk_attn_bwd",
          lambda: ops.attn_bwd(q, k, v, o, do, lse, Hh, B=Bq, T=T, drop_p=0.5, seed=1, stream_id=0), 2.5 * full)


if __name__ == "__main__":
    main()
