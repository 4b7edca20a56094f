#!/usr/bin/env python3
"""VQ lookup lab: in-stream time per launch of melgpt_vq_argmin_fwd_ex (indices only), `reps` launches back to back
between two HIP events so that host launch overhead is not what is measured; batch sweep, both lanes.
Prints one JSON line per case plus a checksum of the indices so that kernel versions can be compared bit for bit.
(The variants compared in profiles/r01_m_vq_lab.jsonl - selected then by MELGPT_VQ_VAR: 0 register prefetch, 1 reload in
place [kept], 2 scheduling barriers, 3/4/5 coalesced loads through wave-private LDS with 16/12/8 waves, 6 loads only,
7 compute only - are in the tree of commit 740ff5a.)"""
import ctypes
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from melspec_gpt_vqvae_amd import _ffi

DEV = "cuda:0"


def run(z, cb, idx, reps, image=None):
    """image = None: melgpt_vq_argmin_fwd_ex;  (buf, with_lo, fused): melgpt_vq_lookup_image on a prepared image"""
    N, D = z.shape
    grid = ctypes.c_int(0)

    def launch():
        if image is not None:
            _ffi.call("melgpt_vq_lookup_image", _ffi.ptr(z), N, D, N, N * D, D, 1, _ffi.ptr(image[0]), image[1], image[2],
                      _ffi.ptr(idx), None, _ffi.stream())
            return
        _ffi.call("melgpt_vq_argmin_fwd_ex", _ffi.ptr(z), _ffi.dtype_code(z.dtype), N, D, N, N * D, D, 1, _ffi.ptr(cb), 128,
                  _ffi.ptr(idx), None, None, None, None, ctypes.addressof(grid), _ffi.stream())

    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    graph = None
    if os.environ.get("VQ_LAB_GRAPH"):  # the `reps` launches captured once and replayed: no host code between them
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            launch()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(reps):
                launch()
        graph.replay()
        torch.cuda.synchronize()
    best = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        if graph is not None:
            graph.replay()
        else:
            for _ in range(reps):
                launch()
        e.record()
        torch.cuda.synchronize()
        best.append(s.elapsed_time(e) / reps)
    best.sort()
    return best[len(best) // 2], grid.value


def main():
    var = os.environ.get("MELGPT_VQ_VAR", "0")
    # optional: batch sizes as arguments (one size per profiler run keeps the per-kernel stats unmixed)
    batches = tuple(int(a) for a in sys.argv[1:]) or (64, 256, 1024, 4096)
    g = torch.Generator(device="cpu").manual_seed(5)
    cb = torch.randn(128, 256, generator=g).to(DEV)
    if os.environ.get("VQ_LAB_IMAGE"):  # prepared images: plain | fused hi | fused hi+lo (quant_conv folded in)
        from melspec_gpt_vqvae_amd.vqvae.quantizer import CodebookImage

        W = (torch.randn(256, 256, generator=g) / 16).to(DEV)
        bias = (0.1 * torch.randn(256, generator=g)).to(DEV)
        for name, kw in (("image_plain", {}), ("image_fused_hi", dict(conv_weight=W, conv_bias=bias, with_lo=False)),
                         ("image_fused_hi_lo", dict(conv_weight=W, conv_bias=bias, with_lo=True))):
            im = CodebookImage()
            buf = im.get(cb, **kw)
            for B in batches:
                n = B * 265
                z = torch.randn(n, 256, generator=torch.Generator(device="cpu").manual_seed(6 + B)).to(DEV).to(torch.bfloat16)
                idx = torch.empty(n, dtype=torch.int64, device=DEV)
                ms, _ = run(z, cb, idx, 50 if B <= 1024 else 20, image=(buf, im.with_lo, im.fused))
                by = n * (256 * 2 + 8) + (131072 if name != "image_fused_hi_lo" else 2 * 65536) // (1 if name == "image_fused_hi_lo" else 2) + 512
                h = hashlib.sha1(idx.cpu().numpy().tobytes()).hexdigest()[:12]
                print(json.dumps(dict(kernel=f"vq_{name}", graph=bool(os.environ.get("VQ_LAB_GRAPH")), batch=B, vectors=n,
                                      us=round(ms * 1e3, 2), algorithmic_MB=round(by / 1e6, 2), GBps=round(by / ms / 1e6, 1),
                                      frac_hbm=round(by / ms / 1e6 / 8000.0, 4), idx_sha=h)), flush=True)
        return
    for dt, es in ((torch.bfloat16, 2), (torch.float32, 4)):
        for B in batches:
            n = B * 265
            z = torch.randn(n, 256, generator=g).to(DEV).to(dt)
            idx = torch.empty(n, dtype=torch.int64, device=DEV)
            ms, grid = run(z, cb, idx, 50 if B <= 1024 else 20)
            by = n * (256 * es + 8) + 128 * 256 * 4
            h = hashlib.sha1(idx.cpu().numpy().tobytes()).hexdigest()[:12]
            print(json.dumps(dict(kernel=f"vq_argmin_{'bf16' if es == 2 else 'f32'}", var=var, graph=bool(os.environ.get("VQ_LAB_GRAPH")), batch=B, vectors=n, grid=grid,
                                  us=round(ms * 1e3, 2), algorithmic_MB=round(by / 1e6, 2), GBps=round(by / ms / 1e6, 1),
                                  frac_hbm=round(by / ms / 1e6 / 8000.0, 4), idx_sha=h)), flush=True)
        if os.environ.get("VQ_LAB_BF16_ONLY"):
            break


if __name__ == "__main__":
    main()
