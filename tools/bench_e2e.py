#!/usr/bin/env python3
"""End-to-end generation chain (BASELINE.json configs[4]), 16-bit lane (bf16, or fp16 with
`--dtype fp16`), random-init weights, synthetic audio:

    raw wav (22 050 Hz, 10 s) -> HIP STFT / log-mel tile -> VQ-VAE encode + 128-code argmin -> class-GPT samples 265
    codes (KV-cached, one HIP graph replayed per token) -> VQ-VAE decode -> mel (80 x 848) -> MelGAN generator -> wav

The reference runs these stages as separate scripts (feature_extraction/extract_mel_spectrogram.py:193-211 - a
`Pool.map` over the files -, extract_codes.py:63-120, Lit_minGPT.sample + decode_to_img in
callbacks/GPT_callbacks.py:83-105).  One JSON line per batch size: per-stage milliseconds, latency percentiles per
batch and clips per second.

  python tools/bench_e2e.py [--dtype fp16] [--batches 1,16,64,128] [--gpus N]

`--gpus N`: the chain shards per clip with no exchange step, so N ranks (one process per GPU, started here
before anything touches a GPU) each take clips r::N of the global list of N x batch clips, run the same loop with NO
collective on the data path, write their line to a scratch directory, and this parent prints one line per batch size with
the ranks' clips/s SUMMED and the slowest rank's latency.  MELGPT_BENCH_SHARE_GPU=1 lets the ranks share cuda:0 (control-flow
rehearsal on a 1-GPU box; flagged in the output)."""
import argparse
import json
import os
import sys
import tempfile
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
warnings.filterwarnings("ignore")

NAMES = ["mel_frontend", "vq_encode", "gpt_sample_265", "vq_decode", "vocoder"]
REPS = {1: 12, 16: 4, 64: 3}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--batches", default="1,16,64,128", help="clips per GPU per pass, comma separated")
    ap.add_argument("--gpus", type=int, default=1)
    return ap.parse_args()


def chain(a, rank, world, share):
    """one rank's loop; returns the list of per-batch-size records"""
    if a.dtype == "fp16":
        os.environ["MELGPT_HALF"] = "fp16"   # the library's IEEE-half flavour (BASELINE configs[4] names fp16); before the imports
    import torch

    import synth
    from melspec_gpt_vqvae_amd import _ffi
    from melspec_gpt_vqvae_amd.feature_extraction.extract_mel_spectrogram import TRANSFORMS
    from melspec_gpt_vqvae_amd.transformer.minGPT import Lit_minGPT, set_compute_dtype
    from melspec_gpt_vqvae_amd.vocoder import Generator
    from melspec_gpt_vqvae_amd.vqvae import big_model_attn_gan as vq

    dev_index = 0 if share else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(dev_index)
    DEV = f"cuda:{dev_index}"
    HALF = _ffi.HALF_DTYPE  # torch.bfloat16, or torch.float16 under --dtype fp16

    torch.manual_seed(0)     # every rank holds the same (random-init) models
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

    out = []
    for B in [int(x) for x in a.batches.split(",")]:
        reps = REPS.get(B, 3)
        # the global list holds world x B clips; this rank takes clips rank, rank + world, ... (synthetic audio, drawn
        # from a generator seeded by the rank: every rank's clips differ)
        g = torch.Generator().manual_seed(1234 + rank)
        idx = list(range(rank, world * B, world))
        wav = (0.1 * torch.randn(B, 220500, generator=g)).to(DEV)
        c = torch.randint(0, 8, (B, 1), generator=g).to(DEV)
        run(wav, c)  # warm-up (allocations, graph capture)
        stages, totals = [], []
        for _ in range(reps):
            st, seq, audio = run(wav, c)
            stages.append(st)
            totals.append(sum(st))
        totals.sort()
        med = [sorted(s[i] for s in stages)[len(stages) // 2] for i in range(len(NAMES))]
        p50 = totals[len(totals) // 2]
        p95 = totals[min(len(totals) - 1, int(round(0.95 * (len(totals) - 1))))]
        out.append({
            "bench": f"wav -> mel -> VQ encode -> GPT sample 265 -> VQ decode -> MelGAN, {_ffi.HALF}, one MI355X",
            "batch": B, "rank": rank, "world": world, "clip_ids": [idx[0], idx[-1]],
            "stage_ms_median": {n: round(v, 2) for n, v in zip(NAMES, med)},
            "latency_ms": {"p50": round(p50, 1), "p95": round(p95, 1), "runs": reps},
            "clips_per_s": round(B / (p50 * 1e-3), 2),
            "x_realtime": round(B * 10.0 / (p50 * 1e-3), 1),
            "shapes": {"codes": list(seq.shape), "audio": list(audio.shape)}})
        if world == 1:
            print(json.dumps(out[-1]), flush=True)
    return out


def main():
    a = parse()
    share = os.environ.get("MELGPT_BENCH_SHARE_GPU") == "1"
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from melspec_gpt_vqvae_amd.launch import spawn_ranks

        with tempfile.TemporaryDirectory() as d:
            rc = spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], a.gpus, share_gpu=share,
                             env_extra={"MELGPT_E2E_RESULT_DIR": d})
            if rc != 0:
                raise SystemExit(rc)
            ranks = [json.load(open(os.path.join(d, f"rank{r}.json"))) for r in range(a.gpus)]
        for i in range(len(ranks[0])):
            rows = [r[i] for r in ranks]
            line = {"bench": rows[0]["bench"].replace("one MI355X", f"{a.gpus} ranks, clips sharded r::{a.gpus}, no collective"),
                    "batch_per_gpu": rows[0]["batch"], "n_gpus": a.gpus, "clips_per_s": round(sum(r["clips_per_s"] for r in rows), 2),
                    "clips_per_s_by_rank": [r["clips_per_s"] for r in rows],
                    "latency_ms": {"p50": max(r["latency_ms"]["p50"] for r in rows), "p95": max(r["latency_ms"]["p95"] for r in rows)},
                    "stage_ms_median_rank0": rows[0]["stage_ms_median"], "clip_ids_by_rank": [r["clip_ids"] for r in rows]}
            if share:
                line["INVALID_debug_shared_gpu"] = True
            print(json.dumps(line), flush=True)
        return
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    out = chain(a, rank, world, share)
    d = os.environ.get("MELGPT_E2E_RESULT_DIR")
    if d:
        with open(os.path.join(d, f"rank{rank}.json"), "w") as f:
            json.dump(out, f)


if __name__ == "__main__":
    main()
