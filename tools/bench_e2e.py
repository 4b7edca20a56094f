#!/usr/bin/env python3
"""End-to-end generation chain on one GPU (BASELINE.json configs[4]), 16-bit lane (bf16, or fp16 with `--dtype fp16`), random-init
weights, synthetic audio:

    raw wav (22 050 Hz, 10 s) -> HIP STFT / log-mel tile -> VQ-VAE encode + 128-code argmin -> class-GPT samples 265
    codes (KV-cached, one HIP graph replayed per token) -> VQ-VAE decode -> mel (80 x 848) -> MelGAN generator -> wav

The reference runs these stages as separate scripts (feature_extraction/extract_mel_spectrogram.py,
extract_codes.py, Lit_minGPT.sample + decode_to_img in callbacks/GPT_callbacks.py:83-105).  One JSON line per batch
size: per-stage milliseconds, latency percentiles per batch and clips per second."""
import json
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
warnings.filterwarnings("ignore")
if "--dtype" in sys.argv and sys.argv[sys.argv.index("--dtype") + 1] == "fp16":
    os.environ["MELGPT_HALF"] = "fp16"   # the library's IEEE-half flavour (BASELINE configs[4] names fp16); before the imports
import torch

import synth
from melspec_gpt_vqvae_amd.feature_extraction.extract_mel_spectrogram import TRANSFORMS
from melspec_gpt_vqvae_amd.transformer.minGPT import Lit_minGPT, set_compute_dtype
from melspec_gpt_vqvae_amd.vocoder import Generator
from melspec_gpt_vqvae_amd.vqvae import big_model_attn_gan as vq

from melspec_gpt_vqvae_amd import _ffi

DEV = "cuda:0"
HALF = _ffi.HALF_DTYPE  # torch.bfloat16, or torch.float16 under --dtype fp16


def main():
    torch.manual_seed(0)
    vqvae = vq.LitVQVAE(num_embeddings=128, embedding_dim=256).to(DEV).eval()
    vq.set_compute_dtype(vqvae, HALF)
    args = synth.gpt_args(n_layer=24, n_head=16, n_embd=1024, reconstruct_spec="", device=DEV, batch_size=2,
                          learning_rate=1e-6)  # config/config_GPT_vas.py:1-18
    lit = Lit_minGPT(args).to(DEV).eval()
    set_compute_dtype(lit.transformer, HALF)
    lit.first_stage_model = vqvae
    voc = Generator(80, 32, 3).to(DEV).eval()
    for m in voc.modules():
        object.__setattr__(m, "compute_dtype", HALF)

    def sync():
        torch.cuda.synchronize()
        return time.perf_counter()

    @torch.no_grad()
    def run(wav, c):
        B = wav.shape[0]
        t = [sync()]
        _, tile = TRANSFORMS.run(wav, want_mel=False, tile_dtype=HALF)        # (B,1,80,848) in [-1,1]
        t.append(sync())
        codes = vqvae.encode_to_codes(tile)                                            # (B,5,53) int64
        seq = lit.code_reader(codes.reshape(B, -1))                                    # time-major (B,265)
        t.append(sync())
        x0 = torch.zeros(B, 0, dtype=torch.int64, device=DEV)
        new, _ = lit.sample(x0, c, steps=265, sample=True, top_k=64)                   # (B,265) sampled codes
        t.append(sync())
        mel = lit.decode_to_img(new, (B, 256, 5, 53))                                  # (B,1,80,848)
        t.append(sync())
        audio = voc(((mel[:, 0].float() + 1) * 0.5))                                   # (B,1,217088)
        t.append(sync())
        return [1e3 * (b - a) for a, b in zip(t[:-1], t[1:])], seq, audio

    names = ["mel_frontend", "vq_encode", "gpt_sample_265", "vq_decode", "vocoder"]
    for B, reps in ((1, 12), (16, 4), (64, 3)):
        wav = 0.1 * torch.randn(B, 220500, device=DEV)
        c = torch.randint(0, 8, (B, 1), device=DEV)
        run(wav, c)  # warm-up (allocations, graph capture)
        stages, totals = [], []
        for _ in range(reps):
            st, seq, audio = run(wav, c)
            stages.append(st)
            totals.append(sum(st))
        totals.sort()
        med = [sorted(s[i] for s in stages)[len(stages) // 2] for i in range(len(names))]
        p50 = totals[len(totals) // 2]
        p95 = totals[min(len(totals) - 1, int(round(0.95 * (len(totals) - 1))))]
        print(json.dumps({
            "bench": f"wav -> mel -> VQ encode -> GPT sample 265 -> VQ decode -> MelGAN, {_ffi.HALF}, one MI355X", "batch": B,
            "stage_ms_median": {n: round(v, 2) for n, v in zip(names, med)},
            "latency_ms": {"p50": round(p50, 1), "p95": round(p95, 1), "runs": reps},
            "clips_per_s": round(B / (p50 * 1e-3), 2),
            "x_realtime": round(B * 10.0 / (p50 * 1e-3), 1),
            "shapes": {"codes": list(seq.shape), "audio": list(audio.shape)}}), flush=True)


if __name__ == "__main__":
    main()
