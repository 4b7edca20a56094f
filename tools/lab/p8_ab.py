#!/usr/bin/env python3
"""A/B of the persistent GEMM's two K loops on the training step's shapes: melgpt_set_gemm_pingpong(0) (five-slot ring, gemm256.hip)
against (1) (ping-pong over half-tiles, gemm8p.hip; combinations it does not serve fall through to the ring in both arms).
RANDOM operands, arms interleaved in ONE process, median / minimum of per-launch HIP-event times; outputs compared bit for bit."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from melspec_gpt_vqvae_amd import _ffi, ops

DEV = "cuda:0"
M = int(os.environ.get("M", "33920"))
ROUNDS = int(os.environ.get("ROUNDS", "7"))
REPS = int(os.environ.get("REPS", "6"))


def cases():
    g = torch.Generator(device=DEV).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
    x1 = rnd(M, 1024).to(torch.bfloat16)
    x4 = rnd(M, 4096).to(torch.bfloat16)
    res = rnd(M, 1024).to(torch.bfloat16)
    w = {n: (rnd(*s) * 0.02).to(torch.bfloat16) for n, s in
         (("qkv", (3072, 1024)), ("proj", (1024, 1024)), ("fc1", (4096, 1024)), ("fc2", (1024, 4096)))}
    b = {n: rnd(w[n].shape[0]) * 0.02 for n in w}
    pre = torch.empty(M, 4096, device=DEV, dtype=torch.bfloat16)
    if os.environ.get("SQUARES", "1") == "1":
        for n in (4096, 8192):
            sa = (torch.rand(n, n, device=DEV, generator=g) * 2 - 1).to(torch.bfloat16)
            sb = (torch.rand(n, n, device=DEV, generator=g) * 2 - 1).to(torch.bfloat16)
            yield f"square {n}", n, n, n, (lambda sa=sa, sb=sb: ops.gemm(sa, sb))
        del sa, sb
    yield "qkv fwd (bias)", M, 3072, 1024, lambda: ops.gemm(x1, w["qkv"], bias=b["qkv"])
    yield "proj + dropout + residual", M, 1024, 1024, lambda: ops.gemm(x1, w["proj"], bias=b["proj"], residual=res, drop_p=0.5, seed=7, stream_id=3)
    yield "fc1 + GELU + derivative", M, 4096, 1024, lambda: ops.gemm(x1, w["fc1"], bias=b["fc1"], act=ops.ACT_GELU_DACT, pre_out=pre)
    yield "fc2 + dropout + residual", M, 1024, 4096, lambda: ops.gemm(x4, w["fc2"], bias=b["fc2"], residual=res, drop_p=0.5, seed=7, stream_id=5)
    yield "fc1 plain + residual-free", M, 4096, 1024, lambda: ops.gemm(x1, w["fc1"])
    yield "dgrad NN K=4096 (dfc1: dY W)", M, 1024, 4096, lambda: ops.gemm(x4, w["fc1"], b_kmajor=True)
    yield "dgrad NN K=1024 (dfc2)", M, 4096, 1024, lambda: ops.gemm(x1, w["fc2"], b_kmajor=True)
    x3 = torch.cat([x1, x1, x1], 1)
    yield "dgrad NN K=3072 (dqkv)", M, 1024, 3072, lambda: ops.gemm(x3, w["qkv"], b_kmajor=True)
    if os.environ.get("CONVS", "1") == "1":
        for nm, B, H, W, Cin, Cout, stride in (("conv3x3 s1 20x212 256->256", 128, 20, 212, 256, 256, 1), ("conv3x3 s1 5x53 512->512", 128, 5, 53, 512, 512, 1),
                                               ("conv3x3 s2 80x848 128->128", 32, 80, 848, 128, 128, 2), ("conv3x3 s1 10x106 256->256", 128, 10, 106, 256, 256, 1)):
            xc = (rnd(B, H, W, Cin) * 0.5).to(torch.bfloat16)
            wc = (rnd(Cout, 3, 3, Cin) * 0.05).to(torch.bfloat16)
            bc = rnd(Cout) * 0.1
            oh = ((H - 2) // 2 + 0, (W - 2) // 2 + 0) if stride == 2 else (H, W)
            fn = (lambda xc=xc, wc=wc, bc=bc, stride=stride, oh=oh: ops.conv2d_nhwc(xc, wc, bc, stride=stride, pad=(0, 0) if stride == 2 else (1, 1), out_hw=(oh[0] + 1, oh[1] + 1) if stride == 2 else oh))
            yield nm, B * (oh[0] + (1 if stride == 2 else 0)) * (oh[1] + (1 if stride == 2 else 0)), Cout, 9 * Cin, fn
    dmul = rnd(M, 4096).to(torch.bfloat16)
    yield "GELU' dgrad (mul) K=1024", M, 4096, 1024, lambda: ops.gemm(x1, w["fc2"], b_kmajor=True, act=ops.ACT_MUL, residual=dmul)
    for nm, dy, x in (("fc1 4096x1024", x4, x1), ("fc2 1024x4096", x1, x4), ("qkv 3072x1024", x3, x1), ("proj 1024x1024", x1, res)):
        n, k = dy.shape[1], x.shape[1]
        wg = torch.zeros(n, k, device=DEV)
        bg = torch.zeros(n, device=DEV)

        def run(dy=dy, x=x, wg=wg, bg=bg):
            ops.wgrad(dy, x, wg, False, bias_out=bg)
            return torch.cat([wg.flatten(), bg])
        yield "wgrad + bias " + nm, n, k, M, run


def main():
    only = os.environ.get("ONLY")
    for name, m, n, k, fn in cases():
        if only and only not in name:
            continue
        outs, times = {}, {"0": [], "1": []}
        for arm in "01":
            _ffi.lib().melgpt_set_gemm_pingpong(int(arm))
            outs[arm] = fn().clone()
        torch.cuda.synchronize()
        same = bool(torch.equal(outs["0"], outs["1"]))
        nbad = int((outs["0"] != outs["1"]).sum()) if not same else 0
        stable = True
        screen_arm = os.environ.get("SCREEN_ARM", "1")      # which K loop the repeatability screen runs (1: ping-pong, 0: ring)
        _ffi.lib().melgpt_set_gemm_pingpong(int(screen_arm))
        for _ in range(int(os.environ.get("SCREEN", "6"))):
            stable = stable and bool(torch.equal(fn(), outs[screen_arm]))
        for r in range(ROUNDS):
            for arm in ("01" if r % 2 == 0 else "10"):
                _ffi.lib().melgpt_set_gemm_pingpong(int(arm))
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(REPS):
                    fn()
                e.record()
                torch.cuda.synchronize()
                times[arm].append(s.elapsed_time(e) / REPS)
        fl = 2.0 * m * n * k
        rec = {"shape": f"{name} {m}x{n}x{k}", "bit_identical": same, "differing": nbad, "p8_repeatable": stable}
        for arm, tag in (("0", "ring"), ("1", "p8")):
            t = sorted(times[arm])
            rec[tag + "_ms_med"] = round(t[len(t) // 2], 4)
            rec[tag + "_ms_min"] = round(t[0], 4)
            rec[tag + "_tflops_med"] = round(fl / t[len(t) // 2] / 1e9, 1)
        rec["speedup_med"] = round(rec["ring_ms_med"] / rec["p8_ms_med"], 4)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
