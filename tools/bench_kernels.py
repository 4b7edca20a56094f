#!/usr/bin/env python3
"""Per-kernel roofline measurements that complement bench.py (which times the whole step):
  * VQ codebook lookup (melgpt_vq_argmin_fwd, bf16 + f32 lanes) - HBM bound: algorithmic bytes / time vs 8 TB/s,
    batch sweep (BASELINE config 2 is B = 64 -> 16 960 latent vectors = 8.95 MB, i.e. ~1 us at peak: launch-bound)
  * fused attention forward / backward (MFMA bound, 4*T^2*C flop per sequence and layer, x2.5 for the backward)
  * mel frontend (HBM bound: 882 KB in + 275 KB out per 10 s clip)
Prints one JSON object per kernel; HIP-event timing on the launch stream, median of 20 launches after warm-up."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch

from melspec_gpt_vqvae_amd import ops
from melspec_gpt_vqvae_amd.vqvae.quantizer import vq_lookup

DEV = "cuda:0"
HBM_PEAK, MFMA_PEAK = 8000.0, 2500.0  # GB/s, TFLOP/s (dense bf16)
FP32_VALU_PEAK = 157.3  # TFLOP/s, MI355X vector fp32


def med_ms(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    out = []
    cb = torch.randn(128, 256, device=DEV)
    for dt, es in ((torch.bfloat16, 2), (torch.float32, 4)):
        for B in (64, 256, 1024, 4096):
            z = torch.randn(B, 5, 53, 256, device=DEV).to(dt).permute(0, 3, 1, 2)  # channels-last latent
            # `reps` launches between the two events: one launch per event pair measured the host's launch path
            # (~20 us through the Python wrapper), not the 8 us kernel (r01_i rows were taken that way)
            reps = 50 if B <= 1024 else 20

            def many():
                for _ in range(reps):
                    vq_lookup(z, cb, want_quantized=False, want_stats=False)
            ms = med_ms(many, iters=7) / reps
            n = B * 265
            bytes_alg = n * (256 * es + 8) + 128 * 256 * 4
            out.append(dict(kernel=f"vq_argmin_{'bf16' if es == 2 else 'f32'}", batch=B, vectors=n, us=round(ms * 1e3, 2),
                            algorithmic_MB=round(bytes_alg / 1e6, 2), GBps=round(bytes_alg / ms / 1e6, 1),
                            frac_hbm=round(bytes_alg / ms / 1e6 / HBM_PEAK, 4),
                            TFLOPs=round(2.0 * n * 128 * 256 / ms / 1e9, 1)))
    # the quantiser's forward as the module calls it (bf16): indices + quantized tensor + squared-error partials +
    # histogram in the same pass; bytes = read z, write q, write idx
    for B in (64, 1024, 4096):
        z = torch.randn(B, 5, 53, 256, device=DEV).to(torch.bfloat16).permute(0, 3, 1, 2)
        reps = 50 if B <= 1024 else 20

        def many_fwd():
            for _ in range(reps):
                vq_lookup(z, cb)
        ms = med_ms(many_fwd, iters=7) / reps
        n = B * 265
        bytes_alg = n * (256 * 2 * 2 + 8) + 128 * 256 * 4
        out.append(dict(kernel="vq_forward_bf16(idx+quantized+stats)", batch=B, vectors=n, us=round(ms * 1e3, 2),
                        algorithmic_MB=round(bytes_alg / 1e6, 2), GBps=round(bytes_alg / ms / 1e6, 1),
                        frac_hbm=round(bytes_alg / ms / 1e6 / HBM_PEAK, 4)))
    B, H, T = 128, 16, 265
    C = 64 * H
    for dt in (torch.bfloat16,):
        qkv = (0.5 * torch.randn(B * T, 3 * C, device=DEV)).to(dt)
        q, k, v = qkv[:, C:2 * C], qkv[:, :C], qkv[:, 2 * C:]
        for p_drop in (0.0, 0.5):
            ms_f = med_ms(lambda: ops.attn_fwd(q, k, v, H, B=B, T=T, drop_p=p_drop, seed=1, stream_id=0))
            o, lse, _ = ops.attn_fwd(q, k, v, H, B=B, T=T, drop_p=p_drop, seed=1, stream_id=0)
            do = torch.randn_like(o)
            ms_b = med_ms(lambda: ops.attn_bwd(q, k, v, o, do, lse, H, B=B, T=T, drop_p=p_drop, seed=1, stream_id=0))
            full = 4.0 * T * T * C * B
            out.append(dict(kernel="attn_fwd_bf16", dropout=p_drop, B=B, H=H, T=T, us=round(ms_f * 1e3, 1),
                            TFLOPs_full=round(full / ms_f / 1e9, 1), TFLOPs_causal=round(full / 2 / ms_f / 1e9, 1),
                            frac_mfma_full=round(full / ms_f / 1e9 / MFMA_PEAK, 4)))
            out.append(dict(kernel="attn_bwd_bf16", dropout=p_drop, B=B, H=H, T=T, us=round(ms_b * 1e3, 1),
                            TFLOPs_full=round(2.5 * full / ms_b / 1e9, 1),
                            frac_mfma_full=round(2.5 * full / ms_b / 1e9 / MFMA_PEAK, 4)))
    from melspec_gpt_vqvae_amd.feature_extraction.extract_mel_spectrogram import TRANSFORMS

    # the frontend's own bound is fp32 VALU: per frame 2.5 N log2 N (real 1024-point FFT) + 16 per one-sided bin
    # (unpacking, magnitude) + 2 per window sample and per non-zero filter weight (680) ~ 37.2 kFLOP, 862 frames
    flop_clip = 862 * (2.5 * 1024 * 10 + 16 * 513 + 2 * 1024 + 2 * 680)
    for n in (8, 64, 512):
        wav = 0.1 * torch.randn(n, 220500, device=DEV)
        ms = med_ms(lambda: TRANSFORMS.run(wav, tile_dtype=torch.bfloat16))
        bytes_alg = n * (220500 * 4 + 80 * 860 * 4 + 80 * 848 * 2)
        out.append(dict(kernel="mel_frontend", clips=n, us=round(ms * 1e3, 1), clips_per_s=round(n / ms * 1e3, 1),
                        GBps=round(bytes_alg / ms / 1e6, 1), frac_hbm=round(bytes_alg / ms / 1e6 / HBM_PEAK, 4),
                        TFLOPs_fp32=round(n * flop_clip / ms / 1e9, 2),
                        frac_fp32_valu=round(n * flop_clip / ms / 1e9 / FP32_VALU_PEAK, 4)))
    for r in out:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
