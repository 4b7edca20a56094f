"""Nearest-neighbour vector quantiser on the HIP kernels of csrc/vq.hip.

Host-side mirror of `VectorQuantizer` (reference vqvae/big_model_attn_gan.py:8-71): same
constructor, `forward(inputs) -> (loss, quantized, (perplexity, encodings, encoding_indices))`,
`get_codebook_entry(indices, shape)`, parameter name `_embedding.weight`.  The (N,K) distance and
one-hot matrices of the reference are never materialised on the hot path; `encodings` is produced by
a tiny kernel only because it is part of the returned tuple.
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn

from .. import _ffi


def _latent_addressing(x):
    """(B,C,H,W) tensor -> (tensor, N, inner, s_outer, s_inner, s_c) for the C ABI, copying only when the
    strides cannot be expressed as  (n // HW) * s_outer + (n % HW) * s_inner + c * s_c."""
    B, Cc, H, W = x.shape
    sB, sC, sH, sW = x.stride()
    if not (sH == W * sW or H == 1):
        x = x.contiguous()
        sB, sC, sH, sW = x.stride()
    return x, B * H * W, H * W, sB, sW, sC


def vq_lookup(z, codebook, want_quantized=True, want_stats=True, want_distances=False):
    """One launch of melgpt_vq_argmin_fwd[_ex].  z: (B,D,H,W) f32/bf16 CUDA tensor (NCHW or channels-last
    strides); codebook (K,D) f32.  Returns dict(indices (N,) int64, quantized, sq_err, histogram, grid, distances)."""
    L = _ffi.lib()
    z, N, inner, s_outer, s_inner, s_c = _latent_addressing(z)
    K, D = codebook.shape
    dev = z.device
    cb = codebook.detach()
    if cb.dtype != torch.float32 or not cb.is_contiguous():
        cb = cb.float().contiguous()
    idx = torch.empty(N, dtype=torch.int64, device=dev)
    q = torch.empty_like(z) if want_quantized else None  # preserves strides (memory format)
    if q is not None and q.stride() != z.stride():
        q = torch.empty_strided(z.shape, z.stride(), dtype=z.dtype, device=dev)
    sq = torch.empty(L.melgpt_vq_max_grid(), dtype=torch.float32, device=dev) if want_stats else None
    hist = torch.zeros(K, dtype=torch.int32, device=dev) if want_stats else None
    dist = torch.empty(N, K, dtype=torch.float32, device=dev) if want_distances else None
    grid = ctypes.c_int(0)
    _ffi.call("melgpt_vq_argmin_fwd_ex", _ffi.ptr(z), _ffi.dtype_code(z.dtype), N, D, inner, s_outer, s_inner,
              s_c, _ffi.ptr(cb), K, _ffi.ptr(idx), _ffi.ptr(q), _ffi.ptr(sq), _ffi.ptr(hist), _ffi.ptr(dist),
              ctypes.addressof(grid), _ffi.stream())
    return dict(indices=idx, quantized=q, sq_err=sq, histogram=hist, grid=grid.value, distances=dist, z=z,
                addressing=(N, inner, s_outer, s_inner, s_c))


class _VQ(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, codebook, commitment_cost):
        r = vq_lookup(inputs, codebook)
        N = r["indices"].numel()
        K, D = codebook.shape
        out = torch.empty(3, dtype=torch.float32, device=inputs.device)
        _ffi.call("melgpt_vq_finalize", _ffi.ptr(r["sq_err"]), r["grid"], _ffi.ptr(r["histogram"]), K, N, D,
                  float(commitment_cost), _ffi.ptr(out), _ffi.stream())
        ctx.save_for_backward(r["z"], codebook, r["indices"])
        ctx.addressing = r["addressing"]
        ctx.commitment = float(commitment_cost)
        idx = r["indices"].unsqueeze(1)
        ctx.mark_non_differentiable(idx)
        loss, perplexity = out[0], out[1].detach()
        return loss, r["quantized"], perplexity, idx

    @staticmethod
    def backward(ctx, g_loss, g_q, _gp, _gi):
        z, codebook, idx = ctx.saved_tensors
        N, inner, s_outer, s_inner, s_c = ctx.addressing
        K, D = codebook.shape
        dz = torch.empty_strided(z.shape, z.stride(), dtype=z.dtype, device=z.device) if ctx.needs_input_grad[0] else None
        dcb = torch.zeros(K, D, dtype=torch.float32, device=z.device) if ctx.needs_input_grad[1] else None
        if g_q is not None:
            if g_q.stride() != z.stride() or g_q.dtype != z.dtype:
                g = torch.empty_strided(z.shape, z.stride(), dtype=z.dtype, device=z.device)
                g.copy_(g_q)
                g_q = g
        gl = g_loss.float().reshape(1).contiguous() if g_loss is not None else None
        cb = codebook.detach().float().contiguous()
        _ffi.call("melgpt_vq_bwd", _ffi.ptr(z), _ffi.ptr(g_q), _ffi.dtype_code(z.dtype), N, D, inner, s_outer,
                  s_inner, s_c, _ffi.ptr(cb), K, _ffi.ptr(idx), _ffi.ptr(gl), ctx.commitment, _ffi.ptr(dz),
                  _ffi.ptr(dcb), _ffi.stream())
        if dcb is not None and dcb.dtype != codebook.dtype:
            dcb = dcb.to(codebook.dtype)
        return dz, dcb, None


class VectorQuantizer(nn.Module):
    """Drop-in for big_model_attn_gan.VectorQuantizer (reference :8-71)."""

    def __init__(self, num_embeddings, embedding_dim, commitment_cost):
        super().__init__()
        self._embedding_dim = embedding_dim
        self._num_embeddings = num_embeddings
        self._embedding = nn.Embedding(num_embeddings, embedding_dim)
        self._embedding.weight.data.uniform_(-1 / num_embeddings, 1 / num_embeddings)  # reference :16
        self._commitment_cost = commitment_cost

    def forward(self, inputs):
        loss, quantized, perplexity, idx = _VQ.apply(inputs, self._embedding.weight, self._commitment_cost)
        N = idx.shape[0]
        encodings = torch.empty(N, self._num_embeddings, dtype=torch.float32, device=inputs.device)
        _ffi.call("melgpt_vq_onehot", _ffi.ptr(idx), N, self._num_embeddings, _ffi.ptr(encodings), _ffi.stream())
        return loss, quantized, (perplexity, encodings, idx)

    @torch.no_grad()
    def encode_indices(self, inputs):
        """Hot path of feature_extraction/extract_codes.py:48-50: latents -> (B,H,W) int64 codes, nothing else."""
        r = vq_lookup(inputs, self._embedding.weight, want_quantized=False, want_stats=False)
        B, _, H, W = inputs.shape
        return r["indices"].view(B, H, W)

    def get_codebook_entry(self, indices, shape):
        """reference :56-71; shape = (batch, height, width, channel) or None."""
        cb = self._embedding.weight.detach().float().contiguous()
        K, D = cb.shape
        idx = indices.reshape(-1).to(torch.int64).contiguous()
        N = idx.numel()
        if shape is None:
            out = torch.empty(N, D, dtype=torch.float32, device=idx.device)
            _ffi.call("melgpt_vq_gather", _ffi.ptr(idx), N, _ffi.ptr(cb), K, D, _ffi.ptr(out), _ffi.F32, N, 0, D, 1,
                      _ffi.stream())
            return out
        B, H, W, Cc = shape
        out = torch.empty(B, Cc, H, W, dtype=torch.float32, device=idx.device)  # contiguous NCHW like the reference
        _ffi.call("melgpt_vq_gather", _ffi.ptr(idx), N, _ffi.ptr(cb), K, D, _ffi.ptr(out), _ffi.F32, H * W,
                  Cc * H * W, 1, H * W, _ffi.stream())
        return out
