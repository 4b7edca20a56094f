"""Oracle: VQ-VAE encoder / nearest-neighbour codebook / decoder as pure functions over a
state_dict (torch CPU fp32).  TEST INFRASTRUCTURE - see oracle/__init__.py.

Restates /root/reference/vqvae/big_model_attn_gan.py: VectorQuantizer :19-71,
ResnetBlock :114-135, Normalize :139-140, Downsample :156-162, swish :164-166,
Upsample :182-186, Encoder.forward :254-282, Decoder.forward :361-392, AttnBlock :425-450,
LitVQVAE.encode/decode :604-614, and feature_extraction/extract_codes.py:40-50.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

CH_MULT = (1, 1, 2, 2, 4)  # big_model_attn_gan.py:527
NUM_RES_BLOCKS = 2          # :528
ATTN_LEVEL = 4              # the only level whose width is 53 (:529, :223)


# ------------------------------------------------------------------------------- VQ
def vq_distances(flat, emb):
    """:28-30  d = |x|^2 + |e|^2 - 2 x.e^T, evaluated in this order in fp32."""
    return (torch.sum(flat ** 2, dim=1, keepdim=True) + torch.sum(emb ** 2, dim=1)
            - 2 * torch.matmul(flat, emb.t()))


def vq_forward(z, emb, commitment_cost=0.25):
    """VectorQuantizer.forward :19-54 -> (loss, quantized NCHW, perplexity, encodings, indices (N,1))."""
    x = z.permute(0, 2, 3, 1).contiguous()
    flat = x.view(-1, emb.shape[1])
    idx = torch.argmin(vq_distances(flat, emb), dim=1).unsqueeze(1)
    enc = torch.zeros(idx.shape[0], emb.shape[0], device=idx.device)
    enc.scatter_(1, idx, 1)
    q = torch.matmul(enc, emb).view(x.shape)
    loss = F.mse_loss(q, x.detach()) + commitment_cost * F.mse_loss(q.detach(), x)
    q = x + (q - x).detach()
    avg = enc.mean(dim=0)
    perplexity = torch.exp(-torch.sum(avg * torch.log(avg + 1e-10)))
    return loss, q.permute(0, 3, 1, 2).contiguous(), perplexity, enc, idx


def vq_gather(indices, emb, shape):
    """get_codebook_entry :56-71; shape = (B,H,W,C) -> NCHW."""
    return emb[indices].view(shape).permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------------- conv stack
def _swish(x):
    return x * torch.sigmoid(x)


def _gn(sd, name, x):
    return F.group_norm(x, 32, sd[name + ".weight"], sd[name + ".bias"], eps=1e-6)


def _conv(sd, name, x, stride=1, padding=1):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=padding)


def resnet_block(sd, name, x):
    """:114-135 (temb is None, dropout 0)."""
    h = _conv(sd, name + ".conv1", _swish(_gn(sd, name + ".norm1", x)))
    h = _conv(sd, name + ".conv2", _swish(_gn(sd, name + ".norm2", h)))
    if name + ".nin_shortcut.weight" in sd:
        x = _conv(sd, name + ".nin_shortcut", x, padding=0)
    return x + h


def attn_block(sd, name, x):
    """:425-450 single-head spatial attention, scale c^-1/2, softmax over keys."""
    h = _gn(sd, name + ".norm", x)
    q = _conv(sd, name + ".q", h, padding=0)
    k = _conv(sd, name + ".k", h, padding=0)
    v = _conv(sd, name + ".v", h, padding=0)
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w = torch.softmax(torch.bmm(q, k) * (int(c) ** (-0.5)), dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, name + ".proj_out", h, padding=0)


def encoder_forward(sd, x, prefix="", ch_mult=CH_MULT, num_res_blocks=NUM_RES_BLOCKS, taps=None):
    """Encoder.forward :254-282 ( (B,1,80,848) -> (B,z,5,53) )."""
    p = prefix
    h = _conv(sd, p + "conv_in", x)
    if taps is not None:
        taps["conv_in"] = h
    for lvl in range(len(ch_mult)):
        for b in range(num_res_blocks):
            h = resnet_block(sd, f"{p}down.{lvl}.block.{b}", h)
            if f"{p}down.{lvl}.attn.{b}.norm.weight" in sd:
                h = attn_block(sd, f"{p}down.{lvl}.attn.{b}", h)
            if taps is not None:
                taps[f"down.{lvl}.block.{b}"] = h
        if lvl != len(ch_mult) - 1:
            # Downsample :156-159: pad right/bottom by one, 3x3 stride-2 conv without padding
            h = _conv(sd, f"{p}down.{lvl}.downsample.conv", F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)
    h = resnet_block(sd, p + "mid.block_1", h)
    h = attn_block(sd, p + "mid.attn_1", h)
    h = resnet_block(sd, p + "mid.block_2", h)
    return _conv(sd, p + "conv_out", _swish(_gn(sd, p + "norm_out", h)))


def decoder_forward(sd, z, prefix="", ch_mult=CH_MULT, num_res_blocks=NUM_RES_BLOCKS):
    """Decoder.forward :361-392 ( (B,z,5,53) -> (B,1,80,848) )."""
    p = prefix
    h = _conv(sd, p + "conv_in", z)
    h = resnet_block(sd, p + "mid.block_1", h)
    h = attn_block(sd, p + "mid.attn_1", h)
    h = resnet_block(sd, p + "mid.block_2", h)
    for lvl in reversed(range(len(ch_mult))):
        for b in range(num_res_blocks + 1):
            h = resnet_block(sd, f"{p}up.{lvl}.block.{b}", h)
            if f"{p}up.{lvl}.attn.{b}.norm.weight" in sd:
                h = attn_block(sd, f"{p}up.{lvl}.attn.{b}", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")  # Upsample :182-186
            h = _conv(sd, f"{p}up.{lvl}.upsample.conv", h)
    return _conv(sd, p + "conv_out", _swish(_gn(sd, p + "norm_out", h)))


def vqvae_encode(sd, x, **kw):
    """LitVQVAE.encode :604-608."""
    h = encoder_forward(sd, x, prefix="_encoder.", **kw)
    return _conv(sd, "quant_conv", h, padding=0)


def vqvae_decode(sd, q, **kw):
    """LitVQVAE.decode :610-614."""
    return decoder_forward(sd, _conv(sd, "post_quant_conv", q, padding=0), prefix="_decoder.", **kw)


def mel_to_codes(sd, mel, **kw):
    """extract_codes.get_codes (extract_codes.py:40-50): (B,80,860) in [0,1] -> (B,5,53) int64.
    CenterCrop(80,848) of a width-860 image keeps columns [6:854] (albumentations 1.2.1)."""
    x = (2 * mel[:, :, 6:854] - 1).unsqueeze(1)
    z = vqvae_encode(sd, x, **kw)
    _, q, _, _, idx = vq_forward(z, sd["_vq_vae._embedding.weight"])
    return idx.reshape(z.shape[0], q.shape[2], q.shape[3]), z
