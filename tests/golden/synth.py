"""Deterministic synthetic inputs / weights shared by the golden-vector generator
(`make_golden.py`, runs only where /root/reference exists) and by the parity tests
(run anywhere, incl. the GPU box).  Everything is drawn from numpy's legacy
`RandomState`, whose streams are stable across numpy versions, so the big tensors
(weights) never have to be stored in the fixtures - only seeds and outputs are.

This file is test infrastructure; the product package never imports it.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np

SEED = 783435  # the reference's fixed seed (GPT_train.py:56-61)


def rs(seed: int) -> np.random.RandomState:
    return np.random.RandomState(seed & 0x7FFFFFFF)


def normal(seed, shape, std=1.0, mean=0.0, dtype=np.float32):
    return (rs(seed).standard_normal(size=shape) * std + mean).astype(dtype)


def uniform(seed, shape, lo, hi, dtype=np.float32):
    return rs(seed).uniform(lo, hi, size=shape).astype(dtype)


def randint(seed, lo, hi, shape):
    return rs(seed).randint(lo, hi, size=shape).astype(np.int64)


# --------------------------------------------------------------------------- GPT
def gpt_args(vocab_size=128, block_size=266, n_layer=2, n_head=4, n_embd=256, class_size=8,
             embd_pdrop=0.0, resid_pdrop=0.0, attn_pdrop=0.0, n_unmasked=0, last_linear=None,
             **extra):
    """Namespace with the fields the reference's GPT/GPTClass constructors read
    (config/config_GPT_vas.py:1-18)."""
    return SimpleNamespace(vocab_size=vocab_size, block_size=block_size, n_layer=n_layer,
                           n_head=n_head, n_embd=n_embd, class_size=class_size,
                           embd_pdrop=embd_pdrop, resid_pdrop=resid_pdrop, attn_pdrop=attn_pdrop,
                           n_unmasked=n_unmasked, last_linear=last_linear, **extra)


def gpt_state_dict(args, seed, block_size=None, with_embedder=True, out_features=None,
                   bias_std=0.02, ln_jitter=0.05):
    """state_dict (numpy) with the reference's key names.  Unlike the
    reference init (zero biases, unit LayerNorm) every tensor is made non-trivial so
    that bias / affine handling is actually exercised by the parity tests."""
    C, L, V = args.n_embd, args.n_layer, args.vocab_size
    bs = block_size if block_size is not None else args.block_size
    out = out_features if out_features is not None else (args.last_linear or V)
    sd = OrderedDict()
    k = [seed * 1000]

    def nxt():
        k[0] += 1
        return k[0]

    sd["pos_emb"] = normal(nxt(), (1, bs, C), 0.02)
    sd["tok_emb.weight"] = normal(nxt(), (V, C), 0.02)
    for i in range(L):
        p = f"blocks.{i}."
        sd[p + "ln1.weight"] = normal(nxt(), (C,), ln_jitter, 1.0)
        sd[p + "ln1.bias"] = normal(nxt(), (C,), ln_jitter)
        sd[p + "ln2.weight"] = normal(nxt(), (C,), ln_jitter, 1.0)
        sd[p + "ln2.bias"] = normal(nxt(), (C,), ln_jitter)
        for nm in ("key", "query", "value", "proj"):
            sd[p + f"attn.{nm}.weight"] = normal(nxt(), (C, C), 0.02)
            sd[p + f"attn.{nm}.bias"] = normal(nxt(), (C,), bias_std)
        sd[p + "mlp.0.weight"] = normal(nxt(), (4 * C, C), 0.02)
        sd[p + "mlp.0.bias"] = normal(nxt(), (4 * C,), bias_std)
        sd[p + "mlp.2.weight"] = normal(nxt(), (C, 4 * C), 0.02)
        sd[p + "mlp.2.bias"] = normal(nxt(), (C,), bias_std)
    sd["ln_f.weight"] = normal(nxt(), (C,), ln_jitter, 1.0)
    sd["ln_f.bias"] = normal(nxt(), (C,), ln_jitter)
    sd["head.weight"] = normal(nxt(), (out, C), 0.02)
    if with_embedder:
        sd["embedder.weight"] = normal(nxt(), (args.class_size, C), 1.0)
    return sd


def causal_mask(bs, n_unmasked=0):
    m = np.tril(np.ones((bs, bs), dtype=np.float32))
    m[:n_unmasked, :n_unmasked] = 1
    return m.reshape(1, 1, bs, bs)


# ------------------------------------------------------------------------- VQVAE
VQ_HP = dict(ch=128, out_ch=1, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, attn_resolutions=(53,),
             in_channels=1, resolution=848, z_channels=256)


def _conv(sd, name, cout, cin, k, seed, gain=1.0):
    bound = gain / math.sqrt(cin * k * k)
    sd[name + ".weight"] = uniform(seed, (cout, cin, k, k), -bound, bound)
    sd[name + ".bias"] = uniform(seed + 1, (cout,), -bound, bound)


def _gn(sd, name, c, seed):
    sd[name + ".weight"] = normal(seed, (c,), 0.05, 1.0)
    sd[name + ".bias"] = normal(seed + 1, (c,), 0.05)


def _res(sd, name, cin, cout, seed):
    _gn(sd, name + ".norm1", cin, seed)
    _conv(sd, name + ".conv1", cout, cin, 3, seed + 2, gain=1.7)
    _gn(sd, name + ".norm2", cout, seed + 4)
    _conv(sd, name + ".conv2", cout, cout, 3, seed + 6, gain=1.7)
    if cin != cout:
        _conv(sd, name + ".nin_shortcut", cout, cin, 1, seed + 8)


def resblock_state_dict(seed, c):
    """one ResnetBlock (c -> c channels), keys as in the module (norm1, conv1, norm2, conv2): the gradient fixture's weights"""
    sd = OrderedDict()
    _res(sd, "blk", c, c, seed)
    return OrderedDict((k[len("blk."):], v) for k, v in sd.items())


def _attn(sd, name, c, seed):
    _gn(sd, name + ".norm", c, seed)
    for j, nm in enumerate(("q", "k", "v", "proj_out")):
        _conv(sd, f"{name}.{nm}", c, c, 1, seed + 2 + 2 * j)


def encoder_state_dict(seed, ch=128, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, z_channels=256,
                       in_channels=1, attn_levels=(4,)):
    """Keys of `Encoder` (big_model_attn_gan.py:190-251).  torch-default-like uniform
    conv init (slightly hotter so activations keep O(1) scale through 20+ layers)."""
    sd = OrderedDict()
    s = [seed * 100000]

    def nxt(n=16):
        s[0] += n
        return s[0]

    _conv(sd, "conv_in", ch, in_channels, 3, nxt())
    in_mult = (1,) + tuple(ch_mult)
    block_in = ch
    for lvl in range(len(ch_mult)):
        block_in = ch * in_mult[lvl]
        block_out = ch * ch_mult[lvl]
        for b in range(num_res_blocks):
            _res(sd, f"down.{lvl}.block.{b}", block_in, block_out, nxt())
            block_in = block_out
            if lvl in attn_levels:
                _attn(sd, f"down.{lvl}.attn.{b}", block_in, nxt())
        if lvl != len(ch_mult) - 1:
            _conv(sd, f"down.{lvl}.downsample.conv", block_in, block_in, 3, nxt())
    _res(sd, "mid.block_1", block_in, block_in, nxt())
    _attn(sd, "mid.attn_1", block_in, nxt())
    _res(sd, "mid.block_2", block_in, block_in, nxt())
    _gn(sd, "norm_out", block_in, nxt())
    _conv(sd, "conv_out", z_channels, block_in, 3, nxt())
    return sd


def decoder_state_dict(seed, ch=128, out_ch=1, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2,
                       z_channels=256, attn_levels=(4,)):
    """Keys of `Decoder` (big_model_attn_gan.py:291-359)."""
    sd = OrderedDict()
    s = [seed * 100000 + 50000]

    def nxt(n=16):
        s[0] += n
        return s[0]

    nl = len(ch_mult)
    block_in = ch * ch_mult[nl - 1]
    _conv(sd, "conv_in", block_in, z_channels, 3, nxt())
    _res(sd, "mid.block_1", block_in, block_in, nxt())
    _attn(sd, "mid.attn_1", block_in, nxt())
    _res(sd, "mid.block_2", block_in, block_in, nxt())
    for lvl in reversed(range(nl)):
        block_out = ch * ch_mult[lvl]
        for b in range(num_res_blocks + 1):
            _res(sd, f"up.{lvl}.block.{b}", block_in, block_out, nxt())
            block_in = block_out
            if lvl in attn_levels:
                _attn(sd, f"up.{lvl}.attn.{b}", block_in, nxt())
        if lvl != 0:
            _conv(sd, f"up.{lvl}.upsample.conv", block_in, block_in, 3, nxt())
    _gn(sd, "norm_out", block_in, nxt())
    _conv(sd, "conv_out", out_ch, block_in, 3, nxt())
    return sd


def vqvae_state_dict(seed, num_embeddings=128, embedding_dim=256, codebook="normal", **hp):
    """Encoder + quant convs + codebook + decoder, keys as in `LitVQVAE`
    (big_model_attn_gan.py:551-579); the GAN discriminator is not on the path."""
    sd = OrderedDict()
    for k, v in encoder_state_dict(seed, **hp).items():
        sd["_encoder." + k] = v
    zc = hp.get("z_channels", 256)
    _conv(sd, "quant_conv", embedding_dim, zc, 1, seed * 100000 + 90000)
    _conv(sd, "post_quant_conv", zc, embedding_dim, 1, seed * 100000 + 90010)
    if codebook == "normal":
        sd["_vq_vae._embedding.weight"] = normal(seed * 100000 + 90020, (num_embeddings, embedding_dim))
    else:  # the reference's default init (big_model_attn_gan.py:16)
        sd["_vq_vae._embedding.weight"] = uniform(seed * 100000 + 90020, (num_embeddings, embedding_dim),
                                                  -1.0 / num_embeddings, 1.0 / num_embeddings)
    dhp = {k: v for k, v in hp.items() if k != "in_channels"}
    for k, v in decoder_state_dict(seed, **dhp).items():
        sd["_decoder." + k] = v
    return sd


_MEL_STATS = None


def mel_tiles(seed, batch, n_mels=80, length=860):
    """Synthetic `*_mel.npy`-like tiles in [0,1]: clip(N(mu_f, sigma_f), 0, 1) with a smooth
    per-bin mean/std profile shaped like data/train_means_stds_melspec_10s_22050hz.txt
    (mean ~0.5 falling with frequency, std ~0.08-0.12).  SURVEY §8d config 1/2."""
    f = np.arange(n_mels, dtype=np.float64) / (n_mels - 1)
    mu = 0.55 - 0.30 * f
    sd = 0.08 + 0.04 * np.sin(np.pi * f)
    x = rs(seed).standard_normal(size=(batch, n_mels, length)) * sd[None, :, None] + mu[None, :, None]
    return np.clip(x, 0.0, 1.0).astype(np.float32)


def mel_matrix(seed=910):
    """synthetic (80, 862) f64 mel-magnitude matrix for the transform tail (M2): columns sweep 1e-8 .. 1e3, with exact
    zeros, a run exactly on LowerThresh's floor and a run exactly on the upper clip edge."""
    m = np.abs(normal(seed, (80, 862), 1.0).astype(np.float64)) * np.logspace(-8, 3, 862)[None, :]
    m[::9, ::5] = 0.0
    m[3, 10:20] = 1e-5
    m[4, 10:20] = 10.0
    return m


def standin_mel_basis(seed=900):
    """NOT librosa's filterbank: a seeded non-negative (80, 513) f32 matrix standing where librosa.filters.mel's result
    stands when the reference's own numpy lines are recorded (make_golden.gen_mel)."""
    return np.abs(normal(seed, (80, 513), 0.05)).astype(np.float32)


def standin_stft(x, hop_length=256, seeds=(901, 902)):
    """NOT an STFT: a seeded complex (513, 1 + len(x)//hop) matrix with librosa's output dtype rule (complex64 for f32
    input, complex128 otherwise), magnitudes sweeping 1e-9 .. 1e2 with exact zeros."""
    n_frames = 1 + len(x) // hop_length
    re_ = normal(seeds[0], (513, n_frames), 1.0).astype(np.float64)
    im_ = normal(seeds[1], (513, n_frames), 1.0).astype(np.float64)
    re_[:, ::7] = 0.0
    im_[:, ::7] = 0.0
    z = (re_ + 1j * im_) * np.logspace(-9, 2, n_frames)[None, :]
    return z.astype(np.complex64 if x.dtype == np.float32 else np.complex128)


def standin_wav(tag, seed=903):
    n = {"short": 100000, "long": 300000, "exact": 220500}[tag]
    return normal(seed, (n,), 0.1).astype(np.float32)


def waveform(seed, n=220500, sr=22050):
    """0.1*N(0,1) + a sine mix."""
    t = np.arange(n, dtype=np.float64) / sr
    y = 0.1 * rs(seed).standard_normal(n)
    for k, f0 in enumerate((220.0, 1000.0, 3300.0)):
        y += (0.3 / (k + 1)) * np.sin(2 * np.pi * f0 * t + 0.1 * k)
    return y


def melgan_state_dict(seed, input_size=80, ngf=32, n_residual_layers=3):
    """state_dict (numpy) of the reference's MelGAN Generator (vocoder/modules.py:38-79) in torch's old-style
    weight_norm parametrisation: per conv `weight_g` (dim0,1,1), `weight_v`, `bias`.  Non-trivial g / bias so that
    the normalisation and bias paths are exercised."""
    sd = OrderedDict()
    k = [seed * 1000]

    def nxt():
        k[0] += 1
        return k[0]

    def wn(name, d0, d1, ks):
        sd[name + ".bias"] = normal(nxt(), (d1 if name.endswith("T") else d0,), 0.05)
        sd[name + ".weight_g"] = uniform(nxt(), (d0, 1, 1), 0.5, 1.5)
        sd[name + ".weight_v"] = normal(nxt(), (d0, d1, ks), 0.1)

    def conv(name, cout, cin, ks):
        sd[name + ".bias"] = normal(nxt(), (cout,), 0.05)
        sd[name + ".weight_g"] = uniform(nxt(), (cout, 1, 1), 0.5, 1.5)
        sd[name + ".weight_v"] = normal(nxt(), (cout, cin, ks), 0.1)

    def convT(name, cin, cout, ks):
        sd[name + ".bias"] = normal(nxt(), (cout,), 0.05)
        sd[name + ".weight_g"] = uniform(nxt(), (cin, 1, 1), 0.5, 1.5)
        sd[name + ".weight_v"] = normal(nxt(), (cin, cout, ks), 0.1)

    ratios = [8, 8, 2, 2]
    mult = 2 ** len(ratios)
    conv("model.1", mult * ngf, input_size, 7)
    i = 2
    for r in ratios:
        convT(f"model.{i + 1}", mult * ngf, mult * ngf // 2, 2 * r)
        i += 2
        for j in range(n_residual_layers):
            dim = mult * ngf // 2
            conv(f"model.{i}.block.2", dim, dim, 3)
            conv(f"model.{i}.block.4", dim, dim, 1)
            conv(f"model.{i}.shortcut", dim, dim, 1)
            i += 1
        mult //= 2
    conv(f"model.{i + 2}", 1, ngf, 7)
    return sd
