#!/usr/bin/env python3
"""VQ-encode alone (BASELINE configs[1]: LitVQVAE encoder + folded quant_conv + 128-code argmin), 16-bit lane:
  python tools/bench_encode.py [--batch 128] [--reps 10]
One JSON line: ms per batch, tiles/s, encoder TFLOP/s on the 142.57 GFLOP of convolutions per tile.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel table of the encoder (tools/profile_encode.sh)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch

import synth
from melspec_gpt_vqvae_amd import _ffi
from melspec_gpt_vqvae_amd.vqvae import big_model_attn_gan as vq


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(783435)
    m = vq.LitVQVAE(num_embeddings=128, embedding_dim=256)
    with torch.no_grad():
        m._vq_vae._embedding.weight.normal_(0.0, 1.0)
    m.to(dev).eval()
    vq.set_compute_dtype(m, _ffi.HALF_DTYPE)
    mel = torch.from_numpy(synth.mel_tiles(783435, a.batch))
    x = (2 * mel[:, :, 6:854] - 1).unsqueeze(1).contiguous().to(dev)
    with torch.no_grad():
        for _ in range(2):
            codes = m.encode_to_codes(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            codes = m.encode_to_codes(x)
        torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / a.reps
    print(json.dumps({"bench": f"VQ-encode + argmin, {a.batch} tiles (1,80,848), {_ffi.HALF}", "ms": round(ms, 3),
                      "tiles_per_s": round(a.batch / ms * 1e3, 1),
                      "encoder_tflops": round(a.batch * 142.57e9 / (ms * 1e-3) / 1e12, 1), "codes": list(codes.shape)}))


if __name__ == "__main__":
    main()
