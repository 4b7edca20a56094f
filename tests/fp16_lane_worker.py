"""Child process of tests/test_fp16_lane_gpu.py (test infrastructure): runs with MELGPT_HALF=fp16, i.e. on
libmelgpt_hip_fp16.so - the same kernels with IEEE half as the 16-bit storage format - and prints one JSON object of
measured errors against fp32 references / goldens recorded from the real reference."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)

import numpy as np
import torch
import torch.nn.functional as F

import synth
from util import golden, rel_err, t

DEV = "cuda:0"


def main():
    from melspec_gpt_vqvae_amd import _ffi, ops

    assert _ffi.HALF == "fp16" and _ffi.HALF_DTYPE == torch.float16 and _ffi.LIB_PATH.endswith("libmelgpt_hip_fp16.so")
    out = {"lib": os.path.basename(_ffi.LIB_PATH)}
    H = torch.float16
    # ---- the other 16-bit format is refused, not silently reinterpreted
    try:
        ops.gemm(torch.zeros(16, 64, device=DEV, dtype=torch.bfloat16), torch.zeros(16, 64, device=DEV, dtype=torch.bfloat16))
        out["bf16_refused"] = False
    except _ffi.MelgptError:
        out["bf16_refused"] = True
    # ---- GEMM: products of fp16 values are exact in f32, accumulation f32 (128-tile and persistent kernels, K-major too)
    torch.manual_seed(1)
    errs = []
    for (M, N, K, bk) in ((300, 1024, 512, False), (33920, 1024, 264, False), (8192, 1024, 320, True)):
        a = (torch.randn(M, K) * 0.5).to(H)
        b = (torch.randn((K, N) if bk else (N, K)) * 0.5).to(H)
        ref = a.float() @ (b.float() if bk else b.float().t())
        got = ops.gemm(a.to(DEV), b.to(DEV), b_kmajor=bk, out_dtype=torch.float32)
        errs.append(rel_err(got.cpu().numpy(), ref.numpy()))
    out["gemm_rel_err_f32_out"] = max(errs)
    a = (torch.randn(16640, 256) * 0.5).to(H)
    b = (torch.randn(1536, 256) * 0.2).to(H)
    bias = torch.randn(1536) * 0.1
    pre = a.float() @ b.float().t() + bias
    dact = torch.empty(16640, 1536, dtype=H, device=DEV)
    act = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), act=ops.ACT_GELU_DACT, pre_out=dact)
    out["gelu_epilogue_rel_err"] = rel_err(act.float().cpu().numpy(), F.gelu(pre).numpy())
    # ---- attention forward / backward against fp32 torch
    B, Hh, T = 2, 4, 265
    C = 64 * Hh
    qkv = (t(synth.normal(70, (B * T, 3 * C)) * np.float32(0.8))).to(H)
    qd = qkv.to(DEV)
    o, lse, _ = ops.attn_fwd(qd[:, :C], qd[:, C:2 * C], qd[:, 2 * C:], Hh, B=B, T=T)
    qr = qkv.float().clone().requires_grad_(True)
    sp = lambda z: z.reshape(B, T, Hh, 64).transpose(1, 2)
    att = (sp(qr[:, :C]) @ sp(qr[:, C:2 * C]).transpose(-2, -1)) / 8.0
    att = F.softmax(att.masked_fill(torch.tril(torch.ones(T, T))[None, None] == 0, float("-inf")), -1)
    yr = (att @ sp(qr[:, 2 * C:])).transpose(1, 2).reshape(B * T, C)
    out["attn_fwd_rel_err"] = rel_err(o.float().cpu().numpy(), yr.detach().numpy())
    do = t(synth.normal(71, (B * T, C))).to(H)
    yr.backward(do.float())
    dq, dk, dv = ops.attn_bwd(qd[:, :C], qd[:, C:2 * C], qd[:, 2 * C:], o, do.to(DEV), lse, Hh, B=B, T=T)
    out["attn_bwd_rel_err"] = max(rel_err(dq.float().cpu().numpy(), qr.grad[:, :C].numpy()),
                                  rel_err(dk.float().cpu().numpy(), qr.grad[:, C:2 * C].numpy()),
                                  rel_err(dv.float().cpu().numpy(), qr.grad[:, 2 * C:].numpy()))
    # ---- 2-layer class-GPT against the real reference's golden (logits, loss), one optimizer step
    from melspec_gpt_vqvae_amd.optim import FusedAdamW
    from melspec_gpt_vqvae_amd.transformer.minGPT import GPTClass, Lit_minGPT, cross_entropy, set_compute_dtype

    g = golden("gptclass_small")
    args = synth.gpt_args(n_layer=2, n_head=4, n_embd=256)
    m = GPTClass(args)
    m.load_state_dict({k: t(v) for k, v in synth.gpt_state_dict(args, int(g["sd_seed"])).items()}, strict=False)
    m.to(DEV).train()
    set_compute_dtype(m, H)
    x, c = t(g["x"], DEV), t(g["c"], DEV)
    logits, _, _ = m(x[:, :-1], c)
    out["gpt_logits_rel_err_vs_reference"] = rel_err(logits.detach().cpu().numpy(), g["logits"])
    loss = cross_entropy(logits.reshape(-1, 128), x.reshape(-1))
    out["gpt_loss_abs_err_vs_reference"] = abs(loss.item() - float(g["loss"]))
    opt = FusedAdamW(m, lr=1e-3)
    opt.zero_grad()
    loss.backward()
    opt.step()
    logits2, _, _ = m(x[:, :-1], c)
    loss2 = cross_entropy(logits2.reshape(-1, 128), x.reshape(-1))
    out["gpt_loss_after_one_step"] = [loss.item(), loss2.item()]
    # ---- the online chain of BASELINE configs[4] in fp16: mel tile -> codes -> greedy sample -> spectrogram
    from melspec_gpt_vqvae_amd.feature_extraction.extract_mel_spectrogram import wav_to_mel
    from melspec_gpt_vqvae_amd.vqvae import big_model_attn_gan as vq

    gv = golden("vqvae_full")
    vae = vq.LitVQVAE(num_embeddings=128, embedding_dim=256)
    vae.load_state_dict({k: t(v) for k, v in synth.vqvae_state_dict(int(gv["seed"])).items()}, strict=False)
    vae.to(DEV).eval()
    vq.set_compute_dtype(vae, H)
    with torch.no_grad():
        z = vae.encode(t(gv["x"], DEV))
        codes = vae.encode_to_codes(t(gv["x"], DEV))
        # the decoder is fed the REFERENCE's codes (gathered from the codebook): one near-tie flipping in the 16-bit
        # encoder swaps a whole 256-vector and shows up as an O(1) error in the image - that is the code-agreement
        # figure above, not a property of the decoder
        emb = vae._vq_vae._embedding.weight.detach()
        q = emb[t(gv["indices"].astype(np.int64), DEV)].reshape(2, 5, 53, 256).permute(0, 3, 1, 2).contiguous()
        rec = vae.decode(q[:1])
    out["vqvae_latent_rel_err_vs_reference"] = rel_err(z.float().cpu().numpy(), gv["z"])
    out["vqvae_code_agreement_vs_reference"] = float((codes.cpu().numpy().ravel() == gv["indices"].astype(np.int64)).mean())
    out["vqvae_rec_rel_err_vs_reference"] = rel_err(rec.float().cpu().numpy(), gv["rec"])
    out["all_finite"] = bool(torch.isfinite(rec).all() and torch.isfinite(z).all())
    wav = [synth.waveform(90).astype(np.float32)]
    mel, tile = wav_to_mel(wav, tile_dtype=H)
    assert tile.dtype == H
    out["mel_tile_fp16_abs_err"] = float((tile[:, 0].float().cpu() - (2 * mel[:, :, 6:854].cpu() - 1)).abs().max())
    gl = golden("lit_mingpt")
    largs = synth.gpt_args(n_layer=2, n_head=4, n_embd=256, reconstruct_spec="", device=DEV, batch_size=2, learning_rate=1e-6)
    lit = Lit_minGPT(largs)
    lit.transformer.load_state_dict({k: t(v) for k, v in synth.gpt_state_dict(largs, int(gl["sd_seed"])).items()}, strict=False)
    lit.to(DEV).eval()
    set_compute_dtype(lit.transformer, H)
    lit.first_stage_model = vae
    batch = {"codes": t(gl["codes"], DEV), "target": t(gl["target"], DEV)}
    xs, _ = lit.sample(lit.get_x(batch)[:, :9], lit.get_c(batch), steps=16, sample=False)
    out["greedy16_token_agreement_vs_reference"] = float((xs.cpu().numpy() == gl["greedy16"]).mean())
    full, _ = lit.sample(lit.get_x(batch)[:, :9], lit.get_c(batch), steps=256, sample=True, top_k=64)
    img = lit.decode_to_img(full, (2, 256, 5, 53))
    out["chain_output_shape"] = list(img.shape)
    out["chain_finite"] = bool(torch.isfinite(img).all())
    print("FP16_LANE_RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
