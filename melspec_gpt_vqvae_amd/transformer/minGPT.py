"""minGPT on the MI355X HIP kernels - host-side mirror of the reference's transformer/minGPT.py.

Same class names, constructor/forward signatures, return tuples, assertions and `state_dict` keys as the
reference (GPTConfig :30-40, CausalSelfAttention :45-90, Block :93-119, GPT :121-199, GPTClass :203-212,
Lit_minGPT :216-665), so it drops in under the reference's training scripts; every tensor op of the forward and
backward pass is a launch through the C ABI (include/melgpt.h).  nn.Linear / nn.LayerNorm / nn.Embedding /
nn.Dropout / nn.GELU children exist only as PARAMETER CONTAINERS that keep the checkpoint ABI - they are never
called.  There is no eager fallback: CPU tensors raise.

Numerics lanes (module attribute `compute_dtype`, see set_compute_dtype):
  torch.float32  - parity lane: exact-f32 MFMA, matches the reference's CPU path to ~1e-6 (gate: 1e-4)
  torch.bfloat16 - throughput lane: bf16 operands/activations, f32 accumulation, f32 master weights & grads
"""
from __future__ import annotations


import numpy as np
import os

import torch
import torch.nn as nn

from .. import _ffi, ops
from ..flat import ensure_flat

try:  # Lightning is optional: the model classes are plain nn.Modules
    import pytorch_lightning as pl

    _LitBase = pl.LightningModule
except Exception:  # pragma: no cover - not installed in the build container
    pl = None
    _LitBase = nn.Module


class GPTConfig:
    """base GPT config, params common to all GPT versions (reference :30-40)"""
    embd_pdrop = 0.1
    resid_pdrop = 0.1
    attn_pdrop = 0.1

    def __init__(self, vocab_size, block_size, **kwargs):
        self.vocab_size = vocab_size
        self.block_size = block_size
        for k, v in kwargs.items():
            setattr(self, k, v)


# ------------------------------------------------------------------------------------------ dropout seeds
class _Seeds:
    """Counter-based dropout: every forward call draws a fresh 64-bit key from (torch.initial_seed(), a
    host-side call counter); kernels derive per-element masks from (key, site id, element index), so the
    backward pass regenerates masks instead of storing them.  No device sync, no generator state on the GPU."""
    counter = 0

    @classmethod
    def next(cls):
        cls.counter += 1
        x = (torch.initial_seed() * 0x9E3779B97F4A7C15 + cls.counter * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 31
        return x & 0x7FFFFFFFFFFFFFFF


def _require_cuda(x):
    if not x.is_cuda:
        raise _ffi.MelgptError("melspec_gpt_vqvae_amd runs on the GPU only: move the module and its inputs to "
                               "'cuda' (there is no CPU / eager fallback)")


def _compute_dtype(module):
    return getattr(module, "compute_dtype", torch.float32)


def set_compute_dtype(module, dtype):
    """torch.float32 (parity lane) or torch.bfloat16 (throughput lane) for `module` and all its children."""
    assert dtype in (torch.float32, _ffi.HALF_DTYPE), f"compute dtype: float32 or {_ffi.HALF_DTYPE} (MELGPT_HALF)"
    for m in module.modules():
        object.__setattr__(m, "compute_dtype", dtype)
    return module


class _BlockWeights:
    """Views into the flat compute buffer / grad buffer for one Block (packed K|Q|V first)."""

    def __init__(self, blk, fp, dtype):
        a, m = blk.attn, blk.mlp
        buf = fp.compute_buffer(dtype)
        self.fp = fp
        self.qkv_p = [a.key.weight, a.query.weight, a.value.weight]
        self.qkvb_p = [a.key.bias, a.query.bias, a.value.bias]
        self.w_qkv = fp.packed(buf, self.qkv_p)
        self.b_qkv = fp.packed(fp.data, self.qkvb_p)
        v = lambda p: fp._slice(buf, p).view(p.shape)
        self.w_proj, self.w_fc1, self.w_fc2 = v(a.proj.weight), v(m[0].weight), v(m[2].weight)
        self.blk = blk


# ========================================================================================== functional core
_FUSE_MASK = os.environ.get("MELGPT_LN_MASK_FUSE", "1") != "0"   # lab switch (A/B of the masked second output)


def _ln_bwd_for_below(owner, dh, x, ln, mu, rs, **kw):
    """LayerNorm backward whose dx is the incoming gradient of the block BELOW `owner` (owner._below): when that block
    replays an MLP dropout mask on it, the masked copy comes out of the same pass and waits on the block
    (`_masked_grad`, keyed by dx's address) - its backward then skips a read + write pass over (B*T, C)."""
    below = getattr(owner, "_below", None)
    m = getattr(below, "_mlp_mask", None) if below is not None else None
    if m is None or not _FUSE_MASK:
        return ops.layernorm_bwd(dh, x, ln.weight, mu, rs, **kw)
    dx, dxm = ops.layernorm_bwd(dh, x, ln.weight, mu, rs, mask=m, **kw)
    # dx itself is kept (so its storage cannot be freed and handed out again at the same address before the consumer
    # looks) together with its version counter (an in-place accumulation into it - a second consumer's gradient added
    # by autograd, a tensor hook - bumps it, and views share their base's counter): a stale masked copy is never taken
    object.__setattr__(below, "_masked_grad", (dx, dx._version, m, dxm))
    return dx


def _take_masked_grad(blk, dy2, mask):
    """the masked copy of dy2 left by the producer of dy2, if it is of exactly this tensor and this mask"""
    hit = getattr(blk, "_masked_grad", None)
    if hit is None:
        return None
    object.__setattr__(blk, "_masked_grad", None)
    dx, version, m, dxm = hit
    if (dx.data_ptr() == dy2.data_ptr() and tuple(dx.shape) == tuple(dy2.shape) and dx.dtype == dy2.dtype
            and dy2._version == version and dx._version == version and m == mask):
        return dxm
    return None


def _attention_core(x2d, w_qkv, b_qkv, w_proj, b_proj, *, B, T, n_head, n_unmasked, attn_p, resid_p, seed, site,
                    residual, want_att):
    """qkv projection -> fused attention -> output projection (+bias, dropout, residual).  x2d (B*T, C)."""
    C = x2d.shape[1]
    qkv = ops.gemm(x2d, w_qkv, bias=b_qkv)                                   # columns [key | query | value]
    k, q, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    a, lse, att = ops.attn_fwd(q, k, v, n_head, B=B, T=T, n_unmasked=n_unmasked, drop_p=attn_p, seed=seed,
                               stream_id=site, want_att=want_att)
    y = ops.gemm(a, w_proj, bias=b_proj, drop_p=resid_p, seed=seed, stream_id=site + 1, residual=residual)
    return y, att, (qkv, a, lse)


# The weight gradients of a Block run on a SECOND stream (round 6; MELGPT_WGRAD_SIDE=0 keeps one stream).  dW = dY^T X is
# independent of the input-gradient chain dX = dY W that the rest of the backward waits for; issued on a side stream its
# workgroups take the CUs the main stream's persistent GEMMs leave idle in their partial last rounds and at their tile
# boundaries (same kernels, same bits: every launch writes its own slice of the flat gradient).  The Block's end joins the
# two streams, in front of the data-parallel hook.  Same box, 20 steps: 93.06 -> 91.66 ms (profiles/r06_f_wgrad_side_ab.jsonl).
# (joined in front of every non-GEMM kernel instead - a weight gradient beside its own input gradient only - the step LOSES:
# 93.65 against 92.7 ms on one stream; what pays is GEMM work beside the VALU-bound attention backward and the HBM-bound
# LayerNorm backward, profiles/r06_h_wgrad_join_ab.jsonl)
WGRAD_SIDE = os.environ.get("MELGPT_WGRAD_SIDE", "1") != "0"
_SIDE = {}


def _wgrad(d, a, gw, acc, gb, accb):
    if not WGRAD_SIDE or not d.is_cuda:
        return ops.wgrad(d, a, gw, acc, bias_out=gb, bias_accumulate=accb)
    main = torch.cuda.current_stream()
    side = _SIDE.get(d.device)
    if side is None:
        side = _SIDE[d.device] = torch.cuda.Stream(device=d.device)
    side.wait_stream(main)                    # d and a are complete where the main stream stands now
    with torch.cuda.stream(side):
        ops.wgrad(d, a, gw, acc, bias_out=gb, bias_accumulate=accb)
    d.record_stream(side)                     # (the allocator must not hand their memory out again while the side stream reads it)
    a.record_stream(side)


def _wgrad_join(device):
    side = _SIDE.get(device) if WGRAD_SIDE else None
    if side is not None:
        torch.cuda.current_stream().wait_stream(side)


def _attention_core_bwd(dy, x2d, saved, w_qkv, w_proj, fp, blkw_params, *, B, T, n_head, n_unmasked, attn_p, resid_p,
                        seed, site, need_dx=True, d_masked=None):
    """backward of _attention_core w.r.t. its input (without the residual path) and its parameters.
    d_masked: dy under the residual dropout's mask, when the producer of dy already wrote it."""
    qkv, a, lse = saved
    C = x2d.shape[1]
    (qkv_p, qkvb_p, proj_w, proj_b) = blkw_params
    if d_masked is not None:
        d = d_masked
    else:
        d = ops.dropout_apply(dy, resid_p, seed, site + 1) if resid_p > 0 else dy   # mask replay
    gb, accb = fp.grad_target(proj_b)
    gw, acc = fp.grad_target(proj_w)
    # dW_proj = d^T a; the bias gradient (column sums of d) rides in the same GEMM's K loop
    _wgrad(d, a, gw, acc, gb, accb)
    da = ops.gemm(d, w_proj, b_kmajor=True)
    dqkv = torch.empty_like(qkv)
    k, q, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    ops.attn_bwd(q, k, v, a, da, lse, n_head, B=B, T=T, dqkv=(dqkv[:, C:2 * C], dqkv[:, :C], dqkv[:, 2 * C:]),
                 n_unmasked=n_unmasked, drop_p=attn_p, seed=seed, stream_id=site)
    gw, acc = fp.packed_grad_target(qkv_p)
    gb, accb = fp.packed_grad_target(qkvb_p)
    _wgrad(dqkv, x2d, gw, acc, gb, accb)    # dW_qkv = dqkv^T x, db_qkv = column sums
    return ops.gemm(dqkv, w_qkv, b_kmajor=True) if need_dx else None


class _BlockFn(torch.autograd.Function):
    """One pre-LN transformer block (reference Block.forward :107-119) = 10 launches forward, 19 backward."""

    @staticmethod
    def forward(ctx, x, blk, want_att, seed, *params):
        B, T, C = x.shape
        dt = _compute_dtype(blk)
        fp = ensure_flat(blk)
        W = _BlockWeights(blk, fp, dt)
        x2 = x.reshape(B * T, C)
        if x2.dtype != dt or not x2.is_contiguous():
            x2 = ops.cast(x2.contiguous(), dt)
        a, m = blk.attn, blk.mlp
        train = blk.training
        attn_p = a.attn_drop.p if train else 0.0
        resid_p = a.resid_drop.p if train else 0.0
        mlp_p = m[3].p if train else 0.0
        site = 16 * getattr(blk, "_layer_index", 0)
        h1, mu1, rs1 = ops.layernorm_fwd(x2, blk.ln1.weight, blk.ln1.bias, blk.ln1.eps)
        x1, att, sav = _attention_core(h1, W.w_qkv, W.b_qkv, W.w_proj, a.proj.bias, B=B, T=T, n_head=a.n_head,
                                       n_unmasked=a.n_unmasked, attn_p=attn_p, resid_p=resid_p, seed=seed, site=site,
                                       residual=x2, want_att=want_att)
        h2, mu2, rs2 = ops.layernorm_fwd(x1, blk.ln2.weight, blk.ln2.bias, blk.ln2.eps)
        # fc1 writes gelu(pre) and gelu'(pre) (one shared exponential); the pre-activation itself is never stored and
        # the backward GEMM's epilogue is a multiplication instead of a second erf evaluation
        dact = torch.empty(B * T, 4 * C, dtype=dt, device=x.device)
        act = ops.gemm(h2, W.w_fc1, bias=m[0].bias, act=ops.ACT_GELU_DACT, pre_out=dact)
        y = ops.gemm(act, W.w_fc2, bias=m[2].bias, drop_p=mlp_p, seed=seed, stream_id=site + 2, residual=x1)
        ctx.blk, ctx.seed, ctx.site = blk, seed, site
        object.__setattr__(blk, "_mlp_mask", (float(mlp_p), int(seed), site + 2) if mlp_p > 0 else None)
        ctx.cfg = (B, T, C, attn_p, resid_p, mlp_p, dt)
        ctx.save_for_backward(x2, mu1, rs1, h1, sav[0], sav[1], sav[2], x1, mu2, rs2, h2, dact, act)
        if att is None:
            att = x.new_zeros(0)
        ctx.mark_non_differentiable(att)
        return y.view(B, T, C), att

    @staticmethod
    def backward(ctx, dy, _datt):
        blk = ctx.blk
        B, T, C, attn_p, resid_p, mlp_p, dt = ctx.cfg
        x2, mu1, rs1, h1, qkv, a_out, lse, x1, mu2, rs2, h2, dact, act = ctx.saved_tensors
        fp = ensure_flat(blk)
        W = _BlockWeights(blk, fp, dt)
        a, m = blk.attn, blk.mlp
        seed, site = ctx.seed, ctx.site
        dy2 = dy.reshape(B * T, C)
        if dy2.dtype != dt or not dy2.is_contiguous():
            dy2 = ops.cast(dy2.contiguous(), dt)
        # ---- MLP branch: y = x1 + drop(fc2(gelu(fc1(ln2(x1)))))
        d = _take_masked_grad(blk, dy2, (float(mlp_p), int(seed), site + 2)) if mlp_p > 0 else dy2
        if d is None:
            d = ops.dropout_apply(dy2, mlp_p, seed, site + 2)
        gb, accb = fp.grad_target(m[2].bias)
        gw, acc = fp.grad_target(m[2].weight)
        _wgrad(d, act, gw, acc, gb, accb)
        dpre = ops.gemm(d, W.w_fc2, b_kmajor=True, act=ops.ACT_MUL, residual=dact)
        gb, accb = fp.grad_target(m[0].bias)
        gw, acc = fp.grad_target(m[0].weight)
        _wgrad(dpre, h2, gw, acc, gb, accb)
        dh2 = ops.gemm(dpre, W.w_fc1, b_kmajor=True)
        g2, accg = fp.grad_target(blk.ln2.weight)
        b2, accb = fp.grad_target(blk.ln2.bias)
        assert accg == accb
        # (dx1 also under the residual dropout's mask, from the same pass, when that dropout is on)
        d1m = None
        if resid_p > 0 and _FUSE_MASK:
            dx1, d1m = ops.layernorm_bwd(dh2, x1, blk.ln2.weight, mu2, rs2, add_in=dy2, dgamma=g2, dbeta=b2,
                                         accumulate=accg, mask=(resid_p, seed, site + 1))
        else:
            dx1 = ops.layernorm_bwd(dh2, x1, blk.ln2.weight, mu2, rs2, add_in=dy2, dgamma=g2, dbeta=b2, accumulate=accg)
        # ---- attention branch: x1 = x + drop(proj(attn(ln1(x))))
        dh1 = _attention_core_bwd(dx1, h1, (qkv, a_out, lse), W.w_qkv, W.w_proj, fp,
                                  (W.qkv_p, W.qkvb_p, a.proj.weight, a.proj.bias), B=B, T=T, n_head=a.n_head,
                                  n_unmasked=a.n_unmasked, attn_p=attn_p, resid_p=resid_p, seed=seed, site=site,
                                  d_masked=d1m)
        g1, accg = fp.grad_target(blk.ln1.weight)
        b1, accb = fp.grad_target(blk.ln1.bias)
        dx = _ln_bwd_for_below(blk, dh1, x2, blk.ln1, mu1, rs1, add_in=dx1, dgamma=g1, dbeta=b1, accumulate=accg)
        _wgrad_join(dx.device)
        hook = getattr(blk, "_grad_ready_hook", None)
        if hook is not None:
            hook(blk)
        return (dx.view(B, T, C), None, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)


class _AttnFn(torch.autograd.Function):
    """Stand-alone CausalSelfAttention.forward (reference :72-90)."""

    @staticmethod
    def forward(ctx, x, mod, seed, *params):
        B, T, C = x.shape
        dt = _compute_dtype(mod)
        fp = ensure_flat(mod)
        buf = fp.compute_buffer(dt)
        qkv_p = [mod.key.weight, mod.query.weight, mod.value.weight]
        qkvb_p = [mod.key.bias, mod.query.bias, mod.value.bias]
        w_qkv, b_qkv = fp.packed(buf, qkv_p), fp.packed(fp.data, qkvb_p)
        w_proj = fp._slice(buf, mod.proj.weight).view(C, C)
        x2 = x.reshape(B * T, C)
        if x2.dtype != dt or not x2.is_contiguous():
            x2 = ops.cast(x2.contiguous(), dt)
        attn_p = mod.attn_drop.p if mod.training else 0.0
        resid_p = mod.resid_drop.p if mod.training else 0.0
        y, att, sav = _attention_core(x2, w_qkv, b_qkv, w_proj, mod.proj.bias, B=B, T=T, n_head=mod.n_head,
                                      n_unmasked=mod.n_unmasked, attn_p=attn_p, resid_p=resid_p, seed=seed, site=0,
                                      residual=None, want_att=True)
        ctx.mod, ctx.seed, ctx.cfg = mod, seed, (B, T, C, attn_p, resid_p, dt)
        ctx.save_for_backward(x2, *sav)
        ctx.mark_non_differentiable(att)
        return y.view(B, T, C), att

    @staticmethod
    def backward(ctx, dy, _datt):
        mod = ctx.mod
        B, T, C, attn_p, resid_p, dt = ctx.cfg
        x2, qkv, a_out, lse = ctx.saved_tensors
        fp = ensure_flat(mod)
        buf = fp.compute_buffer(dt)
        qkv_p = [mod.key.weight, mod.query.weight, mod.value.weight]
        qkvb_p = [mod.key.bias, mod.query.bias, mod.value.bias]
        w_qkv = fp.packed(buf, qkv_p)
        w_proj = fp._slice(buf, mod.proj.weight).view(C, C)
        dy2 = dy.reshape(B * T, C)
        if dy2.dtype != dt or not dy2.is_contiguous():
            dy2 = ops.cast(dy2.contiguous(), dt)
        dx = _attention_core_bwd(dy2, x2, (qkv, a_out, lse), w_qkv, w_proj, fp,
                                 (qkv_p, qkvb_p, mod.proj.weight, mod.proj.bias), B=B, T=T, n_head=mod.n_head,
                                 n_unmasked=mod.n_unmasked, attn_p=attn_p, resid_p=resid_p, seed=ctx.seed, site=0)
        _wgrad_join(dx.device)
        return (dx.view(B, T, C), None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


class CausalSelfAttention(nn.Module):
    """Multi-head masked self-attention with an output projection (reference :45-90).  forward -> (y, att)."""

    def __init__(self, config):
        super().__init__()
        assert config.n_embd % config.n_head == 0
        self.key = nn.Linear(config.n_embd, config.n_embd)
        self.query = nn.Linear(config.n_embd, config.n_embd)
        self.value = nn.Linear(config.n_embd, config.n_embd)
        self.attn_drop = nn.Dropout(config.attn_pdrop)
        self.resid_drop = nn.Dropout(config.resid_pdrop)
        self.proj = nn.Linear(config.n_embd, config.n_embd)
        # checkpoint ABI only: the kernels derive the mask from (row, col, n_unmasked) and never read this buffer
        mask = torch.tril(torch.ones(config.block_size, config.block_size))
        self.n_unmasked = int(getattr(config, "n_unmasked", 0) or 0)
        mask[:self.n_unmasked, :self.n_unmasked] = 1
        self.register_buffer("mask", mask.view(1, 1, config.block_size, config.block_size))
        self.n_head = config.n_head

    def forward(self, x, layer_past=None):
        _require_cuda(x)
        return _AttnFn.apply(x, self, _Seeds.next(), *self.parameters())


class Block(nn.Module):
    """Transformer block; tuple in / tuple out like the reference (:93-119)."""

    def __init__(self, config):
        super().__init__()
        self.ln1 = nn.LayerNorm(config.n_embd)
        self.ln2 = nn.LayerNorm(config.n_embd)
        self.attn = CausalSelfAttention(config)
        self.mlp = nn.Sequential(
            nn.Linear(config.n_embd, 4 * config.n_embd),
            nn.GELU(),
            nn.Linear(4 * config.n_embd, config.n_embd),
            nn.Dropout(config.resid_pdrop),
        )
        self._want_att = True

    def forward(self, x):
        x, _ = x
        _require_cuda(x)
        y, att = _BlockFn.apply(x, self, self._want_att, getattr(self, "_fwd_seed", None) or _Seeds.next(),
                                *self.parameters())
        return y, (att if att.numel() else None)


class _StemFn(torch.autograd.Function):
    """tok_emb(idx) [+ prepended embeddings] + pos_emb -> dropout   (reference GPT.forward :170-180)."""

    @staticmethod
    def forward(ctx, gpt, idx, embeddings, pre_idx, seed, *params):
        dt = _compute_dtype(gpt)
        fp = ensure_flat(gpt)
        B, Tt = idx.shape
        n_pre = 0
        kw = {}
        if pre_idx is not None:
            n_pre = pre_idx.shape[1]
            kw = dict(pre_idx=pre_idx, pre_table=gpt.embedder.weight, n_pre=n_pre)
        elif embeddings is not None:
            n_pre = embeddings.shape[1]
            ev = embeddings
            if ev.dtype != torch.float32 or not ev.is_contiguous():
                ev = ops.cast(ev.contiguous(), torch.float32)
            kw = dict(pre_vals=ev, n_pre=n_pre)
        p = gpt.drop.p if gpt.training else 0.0
        x = ops.embed_fwd(idx, gpt.tok_emb.weight, gpt.pos_emb[0], dtype=dt, drop_p=p, seed=seed, stream_id=0xFFFF0000, **kw)
        ctx.gpt, ctx.cfg = gpt, (n_pre, p, seed, pre_idx is not None, embeddings is not None)
        ctx.save_for_backward(idx, pre_idx if pre_idx is not None else idx.new_zeros(0))
        return x

    @staticmethod
    def backward(ctx, dx):
        gpt = ctx.gpt
        n_pre, p, seed, is_cls, is_vals = ctx.cfg
        idx, pre_idx = ctx.saved_tensors
        fp = ensure_flat(gpt)
        dt = _compute_dtype(gpt)
        if dx.dtype != dt or not dx.is_contiguous():
            dx = ops.cast(dx.contiguous(), dt)
        B, Ttot, C = dx.shape
        tg, acc_t = fp.grad_target(gpt.tok_emb.weight)
        pg_full, acc_p = fp.grad_target(gpt.pos_emb)
        pg = pg_full.view(-1, C)
        if not acc_p:
            ops.zero_(pg[Ttot:])
        assert acc_t == acc_p
        kw = {}
        dvals = None
        if is_cls:
            cg, acc_c = fp.grad_target(gpt.embedder.weight)
            assert acc_c == acc_t
            kw = dict(pre_idx=pre_idx, pre_table_grad=cg)
        elif is_vals:
            dvals = torch.empty(B, n_pre, C, dtype=torch.float32, device=dx.device)
            kw = dict(pre_vals_grad=dvals)
        # token table on the MFMA GEMM (one-hot^T @ dX, split-K); positions / class table / explicit embeddings
        # by the row kernels.  All of them replay the stem's dropout mask.
        ops.embed_table_grad(dx, idx, tg, n_pre=n_pre, accumulate=acc_t, drop_p=p, seed=seed, stream_id=0xFFFF0000)
        ops.embed_bwd(dx, idx, tok_grad=None, pos_grad=pg[:Ttot], n_pre=n_pre, accumulate=acc_t, drop_p=p, seed=seed,
                      stream_id=0xFFFF0000, **kw)
        return (None, None, dvals, None, None) + (None,) * (len(ctx.needs_input_grad) - 5)


class _HeadFn(torch.autograd.Function):
    """ln_f -> head (no bias) -> f32 logits   (reference GPT.forward :186-188)."""

    @staticmethod
    def forward(ctx, x, gpt, *params):
        B, T, C = x.shape
        dt = _compute_dtype(gpt)
        fp = ensure_flat(gpt)
        buf = fp.compute_buffer(dt)
        w = fp._slice(buf, gpt.head.weight).view(gpt.head.weight.shape)
        x2 = x.reshape(B * T, C)
        if x2.dtype != dt or not x2.is_contiguous():
            x2 = ops.cast(x2.contiguous(), dt)
        h, mu, rs = ops.layernorm_fwd(x2, gpt.ln_f.weight, gpt.ln_f.bias, gpt.ln_f.eps)
        logits = ops.gemm(h, w, out_dtype=torch.float32)
        ctx.gpt, ctx.cfg = gpt, (B, T, C, dt)
        ctx.save_for_backward(x2, mu, rs, h)
        return logits.view(B, T, -1)

    @staticmethod
    def backward(ctx, dlogits):
        gpt = ctx.gpt
        B, T, C, dt = ctx.cfg
        x2, mu, rs, h = ctx.saved_tensors
        fp = ensure_flat(gpt)
        buf = fp.compute_buffer(dt)
        w = fp._slice(buf, gpt.head.weight).view(gpt.head.weight.shape)
        d = dlogits.reshape(B * T, -1)
        if d.dtype != dt or not d.is_contiguous():
            d = ops.cast(d.contiguous(), dt)
        gw, acc = fp.grad_target(gpt.head.weight)
        ops.wgrad(d, h, gw, acc)
        dh = ops.gemm(d, w, b_kmajor=True)
        g, accg = fp.grad_target(gpt.ln_f.weight)
        b, accb = fp.grad_target(gpt.ln_f.bias)
        dx = _ln_bwd_for_below(gpt, dh, x2, gpt.ln_f, mu, rs, dgamma=g, dbeta=b, accumulate=accg)
        return (dx.view(B, T, C), None) + (None,) * (len(ctx.needs_input_grad) - 2)


class _CrossEntropyFn(torch.autograd.Function):
    """F.cross_entropy on f32 logits: reduction 'mean' (reference :197, :416) or 'none' (decoders.py:64-68)."""

    @staticmethod
    def forward(ctx, logits2d, target, reduction):
        if not logits2d.is_contiguous():
            logits2d = logits2d.contiguous()
        loss_rows, lse = ops.cross_entropy_fwd(logits2d, target)
        ctx.save_for_backward(logits2d, target, lse)
        ctx.reduction = reduction
        if reduction == "mean":
            return ops.sum_f32(loss_rows, 1.0 / loss_rows.numel()).reshape(())
        return loss_rows

    @staticmethod
    def backward(ctx, g):
        logits, target, lse = ctx.saved_tensors
        g = g.float().contiguous()
        if ctx.reduction == "mean":
            d = ops.cross_entropy_bwd(logits, target, lse, g_scalar=g.reshape(1), g_scale=1.0 / logits.shape[0])
        else:
            d = ops.cross_entropy_bwd(logits, target, lse, g_rows=g.reshape(-1))
        return d, None, None


def cross_entropy(logits, target, reduction="mean"):
    """drop-in for F.cross_entropy(logits (M,V) f32, target (M,)) on the HIP kernels."""
    _require_cuda(logits)
    assert reduction in ("mean", "none")
    return _CrossEntropyFn.apply(logits.float() if logits.dtype != torch.float32 else logits, target, reduction)


class GPT(nn.Module):
    """the full GPT language model, with a context size of block_size (reference :121-199)"""

    def __init__(self, args, embd_pdrop=0., resid_pdrop=0., attn_pdrop=0., n_unmasked=0, last_linear=None,
                 block_size=None):
        super().__init__()
        config = GPTConfig(vocab_size=args.vocab_size, block_size=args.block_size, embd_pdrop=embd_pdrop,
                           resid_pdrop=resid_pdrop, attn_pdrop=attn_pdrop, n_layer=args.n_layer, n_head=args.n_head,
                           n_embd=args.n_embd, n_unmasked=n_unmasked, last_linear=last_linear)
        if block_size is not None:
            config.block_size = block_size
        self.tok_emb = nn.Embedding(config.vocab_size, config.n_embd)
        self.pos_emb = nn.Parameter(torch.zeros(1, config.block_size, config.n_embd))
        self.drop = nn.Dropout(config.embd_pdrop)
        self.blocks = nn.Sequential(*[Block(config) for _ in range(config.n_layer)])
        self.ln_f = nn.LayerNorm(config.n_embd)
        output_size = last_linear if config.last_linear is not None else config.vocab_size
        self.head = nn.Linear(config.n_embd, output_size, bias=False)
        self.block_size = config.block_size
        self.apply(self._init_weights)
        self.config = config
        for i, blk in enumerate(self.blocks):
            object.__setattr__(blk, "_layer_index", i)
            # the module whose backward PRODUCES this block's incoming gradient (next block, or the head) finds the block
            # through `_below` and writes that gradient under the block's MLP dropout mask in the same pass (_masked_grad)
            if i + 1 < len(self.blocks):
                object.__setattr__(self.blocks[i + 1], "_below", blk)
        object.__setattr__(self, "_below", self.blocks[-1] if len(self.blocks) else None)
        self.compute_dtype = torch.float32

    def get_block_size(self):
        return self.block_size

    def _init_weights(self, module):
        # reference :159-166
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=0.02)
            if isinstance(module, nn.Linear) and module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)

    def _trunk(self, idx, embeddings=None, pre_idx=None, want_att=True):
        _require_cuda(idx)
        ensure_flat(self)
        n_pre = embeddings.shape[1] if embeddings is not None else (pre_idx.shape[1] if pre_idx is not None else 0)
        t = idx.shape[1] + n_pre
        assert t <= self.block_size, "Cannot forward, model block size is exhausted."
        seed = _Seeds.next()
        x = _StemFn.apply(self, idx, embeddings, pre_idx, seed, *self._stem_params())
        last = len(self.blocks) - 1
        for i, blk in enumerate(self.blocks):  # only the last block's attention is part of the API (:182-185)
            blk._want_att = want_att and i == last
            object.__setattr__(blk, "_fwd_seed", seed)
        x, att = self.blocks((x, None))
        logits = _HeadFn.apply(x, self, self.ln_f.weight, self.ln_f.bias, self.head.weight)
        return logits, att

    # ------------------------------------------------------------------ KV-cached decoding
    # The reference samples by re-running the whole model on the growing sequence (:293-360, decoders.py:89-123).
    # decode_step feeds ONE position: per layer the new token's k, v are appended to a (B, block_size, C) cache and
    # its query attends to the cached positions (csrc/decode.hip) - same logits as the last row of a full forward.
    def kv_cacheable(self):
        """True when a position's K/V never change once computed, i.e. every block is strictly causal."""
        return all(blk.attn.n_unmasked == 0 for blk in self.blocks)

    @torch.no_grad()
    def decode_begin(self, batch_size):
        """-> a fresh KV cache for `batch_size` sequences (eval mode only)."""
        assert not self.training, "KV-cached decoding is an inference path (dropout is not applied)"
        # positions below n_unmasked attend bidirectionally in the reference (:65-69), so their K/V change as the
        # sequence grows - a cache of them would be stale
        assert self.kv_cacheable(), "KV-cached decoding needs n_unmasked == 0 (use kv_cache=False)"
        dt = _compute_dtype(self)
        dev = self.pos_emb.device
        C = self.tok_emb.weight.shape[1]
        return {"pos": 0, "B": int(batch_size), "pos_dev": torch.zeros(1, dtype=torch.int32, device=dev),
                "k": [torch.empty(batch_size, self.block_size, C, dtype=dt, device=dev) for _ in self.blocks],
                "v": [torch.empty(batch_size, self.block_size, C, dtype=dt, device=dev) for _ in self.blocks]}

    @torch.no_grad()
    def decode_step(self, cache, idx=None, embeddings=None, pre_idx=None, want_att=False):
        """feed one position - a token `idx` (B,1), a class token `pre_idx` (B,1) or an explicit embedding
        `embeddings` (B,1,C) - and return the f32 logits (B, V) for the next one [, last block's attention row
        (B, H, block_size), entries beyond the current position undefined]."""
        assert not self.training
        pos, B = cache["pos"], cache["B"]
        assert pos < self.block_size, "Cannot forward, model block size is exhausted."
        dt = _compute_dtype(self)
        ensure_flat(self)
        C = self.tok_emb.weight.shape[1]
        pe = self.pos_emb[0, pos:pos + 1]
        if idx is not None:
            _require_cuda(idx)
            assert idx.shape == (B, 1)
            x = ops.embed_fwd(idx, self.tok_emb.weight, pe, dtype=dt)
        elif pre_idx is not None:
            assert pre_idx.shape == (B, 1)
            x = ops.embed_fwd(pre_idx.new_zeros(B, 0), self.tok_emb.weight, pe, dtype=dt, pre_idx=pre_idx,
                              pre_table=self.embedder.weight, n_pre=1)
        else:
            assert embeddings is not None and embeddings.shape == (B, 1, C)
            ev = embeddings
            if ev.dtype != torch.float32 or not ev.is_contiguous():
                ev = ops.cast(ev.contiguous(), torch.float32)
            x = ops.embed_fwd(torch.zeros(B, 0, dtype=torch.int64, device=ev.device), self.tok_emb.weight, pe, dtype=dt,
                              pre_vals=ev, n_pre=1)
        logits, att_row = self._decode_trunk(x.view(B, C), cache, pos, None, want_att)
        cache["pos"] = pos + 1
        return (logits, att_row) if want_att else logits

    def _decode_trunk(self, x, cache, pos, pos_dev, want_att):
        """blocks + ln_f + head for one position; x (B, C) in the compute dtype."""
        dt = _compute_dtype(self)
        fp = ensure_flat(self)
        B = x.shape[0]
        att_row = None
        last = len(self.blocks) - 1
        for li, blk in enumerate(self.blocks):
            W = _BlockWeights(blk, fp, dt)
            a, m = blk.attn, blk.mlp
            if want_att and li == last:
                att_row = torch.zeros(B, a.n_head, self.block_size, dtype=torch.float32, device=x.device)
            qkv = ops.linear_rows(x, W.w_qkv, bias=W.b_qkv, ln=(blk.ln1.weight, blk.ln1.bias, blk.ln1.eps))
            y = ops.attn_decode(qkv, cache["k"][li], cache["v"][li], a.n_head, pos,
                                att_row=att_row if li == last else None, pos_dev=pos_dev)
            x1 = ops.linear_rows(y, W.w_proj, bias=a.proj.bias, residual=x)
            act = ops.linear_rows(x1, W.w_fc1, bias=m[0].bias, act=ops.ACT_GELU,
                                  ln=(blk.ln2.weight, blk.ln2.bias, blk.ln2.eps))
            x = ops.linear_rows(act, W.w_fc2, bias=m[2].bias, residual=x1)
        buf = fp.compute_buffer(dt)
        w = fp._slice(buf, self.head.weight).view(self.head.weight.shape)
        return ops.linear_rows(x, w, out_dtype=torch.float32, ln=(self.ln_f.weight, self.ln_f.bias, self.ln_f.eps)), att_row

    @torch.no_grad()
    def decode_sample_graph(self, cache, first_token, steps, *, temperature=1.0, top_k=None, sample=False, seed=0,
                            n_prompt=0):
        """Sample `steps` more tokens after `first_token` (B,1) with ONE captured HIP graph replayed per token: the
        position, the sampling-step number and the token fed back all live on the device (csrc/decode.hip,
        melgpt_sample_logits_dev), so a replay needs no host round trip and ~200 kernel launches cost one graph
        launch.  The cache must already hold positions 0 .. cache['pos']-1; first_token is x[cache['pos']-1].
        -> (B, steps) int64."""
        B, pos0 = cache["B"], cache["pos"]
        assert first_token.shape == (B, 1) and pos0 + steps <= self.block_size
        dt = _compute_dtype(self)
        dev = first_token.device
        idx = first_token.clone().contiguous()
        seq = torch.zeros(B, self.block_size, dtype=torch.int64, device=dev)
        pos_dev = cache["pos_dev"]
        pe = self.pos_emb[0].contiguous()

        def body():
            x = ops.embed_decode(idx, self.tok_emb.weight, pe, pos_dev, dt)
            logits, _ = self._decode_trunk(x, cache, 0, pos_dev, False)
            ops.sample_logits_dev(logits, idx, pos_dev, -n_prompt, temperature=temperature, top_k=top_k, sample=sample,
                                  seed=seed, seq=seq)
            ops.incr_i32(pos_dev)

        if steps <= 0:
            return seq[:, :0]
        # one eager step first: it performs every lazy one-time setup (LDS attributes, bf16 weight shadow, workspaces)
        # outside the capture - and is itself the first of the `steps`
        pos_dev.fill_(pos0)
        body()
        if steps > 1:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                body()                      # second step, on the side stream (warms that stream's workspace)
            torch.cuda.current_stream().wait_stream(side)
        if steps > 2:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()                      # third step is executed by the capture's first replay below
            for _ in range(steps - 2):
                g.replay()
        cache["pos"] = pos0 + steps
        return seq[:, pos0:pos0 + steps].clone()

    def _stem_params(self):
        ps = [self.tok_emb.weight, self.pos_emb]
        if hasattr(self, "embedder"):
            ps.append(self.embedder.weight)
        return ps

    def discard_att(self):
        """Context for callers that ignore forward()'s third result (`logits, _, _ = self.transformer(...)`, reference
        :415, decoders.py:24, encoders.py:33): inside it the last block does not write the (B, H, T, T) f32 attention
        map - 575 MB per step at the VAS training shape - and forward() returns None in its place."""
        return _DiscardAtt(self)

    def forward(self, idx, embeddings=None, targets=None):
        logits, att = self._trunk(idx, embeddings=embeddings, want_att=getattr(self, "_emit_att", True))
        loss = None
        if targets is not None:
            loss = cross_entropy(logits.view(-1, logits.size(-1)), targets.view(-1))
        return logits, loss, att


class _DiscardAtt:
    def __init__(self, gpt):
        self.gpt = gpt

    def __enter__(self):
        self.prev = getattr(self.gpt, "_emit_att", True)
        object.__setattr__(self.gpt, "_emit_att", False)

    def __exit__(self, *exc):
        object.__setattr__(self.gpt, "_emit_att", self.prev)
        return False


class GPTClass(GPT):
    """class-conditioned GPT: embedder(token) is prepended (reference :203-212).  The embedder is created AFTER
    apply(_init_weights), so it keeps nn.Embedding's N(0,1) init, as in the reference."""

    def __init__(self, args):
        super().__init__(args, embd_pdrop=args.embd_pdrop, resid_pdrop=args.resid_pdrop, attn_pdrop=args.attn_pdrop,
                         n_unmasked=args.n_unmasked, last_linear=args.last_linear, block_size=args.block_size)
        self.embedder = nn.Embedding(args.class_size, args.n_embd)

    def forward(self, idx, token):
        logits, att = self._trunk(idx, pre_idx=token, want_att=getattr(self, "_emit_att", True))
        return logits, None, att


# ================================================================================================= per-sequence CE
class _SequenceCEFn(torch.autograd.Function):
    """per-token cross entropy summed per sequence = CrossEntropyLoss(reduction='none') + view + sum(-1)
    (reference decoders.py:64-68).  logits (R*T, V) f32, target (R*T,) -> (R,)"""

    @staticmethod
    def forward(ctx, logits2d, target, T):
        if not logits2d.is_contiguous():
            logits2d = logits2d.contiguous()
        loss_rows, lse = ops.cross_entropy_fwd(logits2d, target)
        ctx.save_for_backward(logits2d, target, lse)
        ctx.T = T
        return ops.group_sum(loss_rows, T)

    @staticmethod
    def backward(ctx, g):
        logits, target, lse = ctx.saved_tensors
        d = ops.cross_entropy_bwd(logits, target, lse, g_rows=g.float().contiguous().reshape(-1), g_group=ctx.T)
        return d, None, None


def sequence_cross_entropy(logits, target):
    """logits (R, T, V), target (R, T) -> (R,) summed token NLL."""
    _require_cuda(logits)
    R, T, V = logits.shape
    lg = logits.float() if logits.dtype != torch.float32 else logits
    return _SequenceCEFn.apply(lg.reshape(R * T, V), target.reshape(-1), T)


def decay_groups(module):
    """The reference's AdamW grouping (minGPT.py:618-665, Lit_GPT_VAE.py:895-943): weights of nn.Linear decay,
    every bias / LayerNorm weight / Embedding weight / pos_emb does not.  Returns (decay names, no_decay names),
    sorted, and asserts the reference's partition invariants (:653-657)."""
    decay, no_decay = set(), set()
    for mn, m in module.named_modules():
        for pn, p in m.named_parameters(recurse=False):
            fpn = "%s.%s" % (mn, pn) if mn else pn
            if pn.endswith("bias"):
                no_decay.add(fpn)
            elif pn.endswith("weight") and isinstance(m, nn.Linear):
                decay.add(fpn)
            elif pn.endswith("weight") and isinstance(m, (nn.LayerNorm, nn.Embedding)):
                no_decay.add(fpn)
            elif pn == "pos_emb":
                no_decay.add(fpn)
    param_dict = {pn: p for pn, p in module.named_parameters()}
    inter, union = decay & no_decay, decay | no_decay
    assert len(inter) == 0, "parameters %s made it into both decay/no_decay sets!" % (str(inter),)
    assert len(param_dict.keys() - union) == 0, "parameters %s were not separated into either decay/no_decay set!" \
        % (str(param_dict.keys() - union),)
    return sorted(decay), sorted(no_decay)


def make_adamw(module, learning_rate, weight_decay=0.01, betas=(0.9, 0.95)):
    decay, no_decay = decay_groups(module)
    pd = {pn: p for pn, p in module.named_parameters()}
    groups = [{"params": [pd[pn] for pn in decay], "weight_decay": weight_decay},
              {"params": [pd[pn] for pn in no_decay], "weight_decay": 0.0}]
    return torch.optim.AdamW(groups, lr=learning_rate, betas=betas)


# ================================================================================================== Lit_minGPT
class Lit_minGPT(_LitBase):
    """Host logic of the reference's LightningModule (minGPT.py:216-665): teacher-forced step, autoregressive
    sampling, VQ-code ordering, optimizer grouping.  A pytorch_lightning.LightningModule when Lightning is
    installed, else a plain nn.Module with the same methods.  Data loading (datasets/, :461-505) and the
    TensorBoard image logging (:530-612) are outside the hot path; `datamodule_loader` is a hook."""

    def __init__(self, args, ckpt_path=None, ignore_keys=[], first_stage_key="image", cond_stage_key="depth",
                 downsample_cond_size=-1, pkeep=1.0):
        super().__init__()
        self.args = args
        self.transformer = GPTClass(args)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)
        self.first_stage_key = first_stage_key
        self.cond_stage_key = cond_stage_key
        self.downsample_cond_size = downsample_cond_size
        self.pkeep = pkeep
        self.datamodule_loader()
        self.forward_shuffle_idx, self.backward_shuffle_idx = self.make_idx(5, 53)
        if getattr(self.args, "reconstruct_spec", "") != "":
            from ..vqvae.big_model_attn_gan import LitVQVAE

            self.first_stage_model = LitVQVAE(num_embeddings=128, embedding_dim=256)
            self.first_stage_model.load_state_dict(torch.load(self.args.reconstruct_spec))
            self.first_stage_model.eval().to(self.args.device)

    def datamodule_loader(self):
        """reference :461-477: DataModule over args.spec_dir_path (80 x 860 spectrograms, centre crop 848, the code
        files beside them).  Modules built without a data path (tests, benches, sampling-only use) get data = None."""
        self.data = None
        if getattr(self.args, "spec_dir_path", None) and getattr(self.args, "load_data", True):
            from ..datasets import DataModule

            kw = {"splits_dir": self.args.splits_dir} if hasattr(self.args, "splits_dir") else {}
            self.data = DataModule(batch_size=self.args.batch_size, spec_dir_path=self.args.spec_dir_path, mel_num=80,
                                   spec_len=860, spec_crop_len=848, random_crop=False,
                                   num_workers=getattr(self.args, "workers", 0), **kw)
            self.data.setup()

    def train_dataloader(self):
        self.len_train_data = len(self.data.train_dataset)
        return self.data.train_dataloader()

    def val_dataloader(self):
        self.len_val_data = len(self.data.val_dataset)
        return self.data.val_dataloader()

    def init_from_ckpt(self, path, ignore_keys=list()):
        sd = torch.load(path, map_location="cpu")["state_dict"]
        for k in list(sd.keys()):
            for ik in ignore_keys:
                if k.startswith(ik):
                    del sd[k]
        self.load_state_dict(sd, strict=False)

    def forward(self, x, c=None):
        """reference :260-285: logits for p(z_i | z_<i, c); the target is the full sequence."""
        with self.transformer.discard_att():
            logits, _, _ = self.transformer(x[:, :-1], c)
        cond_size = c.size(-1)
        return logits[:, cond_size - 1:], x

    def top_k_logits(self, logits, k):
        v, ix = torch.topk(logits, k)
        out = logits.clone()
        out[out < v[..., [-1]]] = -float('Inf')
        return out

    @torch.no_grad()
    def sample(self, x, c, steps, temperature=1.0, sample=False, top_k=None, callback=lambda k: None, kv_cache=True):
        """reference :293-360 (GPTClass branch).  temperature / top-k / softmax / multinomial-or-argmax run in one
        kernel (melgpt_sample_logits).  kv_cache=True (default) feeds one position per step through
        GPT.decode_step; kv_cache=False is the reference's full re-forward per step.  Both return
        (x, last attention on CPU); the cached path runs ONE full forward at the end for that attention map."""
        block_size = self.transformer.get_block_size()
        assert not self.transformer.training
        if self.pkeep <= 0.0:
            raise NotImplementedError('Implement for GPTClass')
        seed = _Seeds.next()
        att = None
        if kv_cache and steps > 0 and self.transformer.kv_cacheable():
            tr = self.transformer
            cond_size = c.size(-1)
            assert cond_size == 1 and x.size(1) + cond_size + steps - 1 <= block_size
            with torch.no_grad():
                cache = tr.decode_begin(x.size(0))
                logits = tr.decode_step(cache, pre_idx=c)
                for j in range(x.size(1)):
                    logits = tr.decode_step(cache, idx=x[:, j:j + 1])
                n_prompt = x.size(1)
                use_graph = steps >= 4 and os.environ.get('MELGPT_DECODE_GRAPH', '1') != '0'
                for k in range(steps):
                    callback(k)
                    ix = ops.sample_logits(logits, temperature=temperature, top_k=top_k, sample=sample, seed=seed, step=k)
                    x = torch.cat((x, ix), dim=1)
                    if use_graph:
                        # the remaining steps run as one captured HIP graph replayed per token (no host round trips)
                        for kk in range(k + 1, steps):
                            callback(kk)
                        rest = tr.decode_sample_graph(cache, ix, steps - 1 - k, temperature=temperature, top_k=top_k,
                                                       sample=sample, seed=seed, n_prompt=n_prompt)
                        x = torch.cat((x, rest), dim=1)
                        break
                    if k + 1 < steps:
                        logits = tr.decode_step(cache, idx=ix)
                _, _, att = tr(x[:, :-1], c)
            return x, att.detach().cpu()
        for k in range(steps):
            callback(k)
            cond_size = c.size(-1)
            assert x.size(1) + cond_size <= block_size
            logits, _, att = self.transformer(x, c)
            last = logits[:, -1, :]
            ix = ops.sample_logits(last, temperature=temperature, top_k=top_k, sample=sample, seed=seed, step=k)
            x = torch.cat((x, ix), dim=1)
        return x, att.detach().cpu()

    def get_x(self, batch):
        """(B,5,53) codes -> (B,265) time-major (:387-394)."""
        x = batch['codes'].to(self.args.device)
        return ops.codes_permute(x, x.shape[1], x.shape[2])

    def get_c(self, batch):
        return batch["target"].unsqueeze(1).to(self.args.device)

    def get_xc(self, batch, N=None):
        x, c = self.get_x(batch), self.get_c(batch)
        if N is not None:
            x, c = x[:N], c[:N]
        return x, c

    def shared_step(self, batch, batch_idx):
        x, c = self.get_xc(batch)
        logits, target = self(x, c)
        return cross_entropy(logits.reshape(-1, logits.size(-1)), target.reshape(-1))

    def training_step(self, batch, batch_idx):
        loss = self.shared_step(batch, batch_idx)
        if pl is not None:
            self.log("train/loss", loss, prog_bar=True, logger=True, on_step=True, on_epoch=True)
        return loss

    def validation_step(self, batch, batch_idx):
        loss = self.shared_step(batch, batch_idx)
        if pl is not None:
            self.log("val/loss", loss, prog_bar=True, logger=True, on_step=True, on_epoch=True)
        return loss

    def make_idx(self, H, W):
        idx = np.arange(H * W).reshape(H, W).T
        idx = torch.tensor(idx.ravel())
        return idx, torch.argsort(idx)

    def code_reader(self, x, reverse=False):
        """reference :438-456 for L == H*W: x (B,265) permuted by the forward / backward shuffle index."""
        B, L = x.shape
        assert L == len(self.forward_shuffle_idx), "only the 5x53 code grid is supported"
        if x.is_cuda:
            return ops.codes_permute(x, 5, 53, reverse=reverse)
        return x[:, self.backward_shuffle_idx if reverse else self.forward_shuffle_idx]

    @torch.no_grad()
    def decode_to_img(self, index, zshape):
        """reference :515-528: codes -> codebook entries -> VQ-VAE decoder."""
        index = self.code_reader(index, reverse=True)
        bhwc = (zshape[0], zshape[2], zshape[3], zshape[1])
        quant_z = self.first_stage_model._vq_vae.get_codebook_entry(index.reshape(-1), shape=bhwc)
        return self.first_stage_model.decode(quant_z)

    def configure_optimizers(self):
        # the reference walks self.transformer only (:632,652): a loaded first_stage_model (VQ-VAE) stays frozen
        return make_adamw(self.transformer, self.args.learning_rate)

    def configure_fused_optimizer(self):
        """MI355X-native alternative: one fused AdamW launch per weight-decay group over the flat store."""
        from ..optim import FusedAdamW

        return FusedAdamW(self.transformer, lr=self.args.learning_rate, betas=(0.9, 0.95), weight_decay=0.01)
