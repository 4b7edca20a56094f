"""Differentiable path through the VQ-VAE (reference: LitVQVAE.forward is differentiable end to end,
vqvae/big_model_attn_gan.py:622-634).

No scored configuration trains the VQ-VAE (the reference's README: it is pre-trained elsewhere), so this path is
CORRECTNESS-first: one `torch.autograd.Function` per layer kind over NHWC tensors, forward on the inference path's own
kernels (unfused: GroupNorm -> convolution), backward on the gradient exports of csrc/vqvae_bwd.hip (both numerics lanes,
deterministic).  The encoder / decoder take it when autograd is recording and something requires a gradient; inference
(torch.no_grad(), frozen module + detached input) keeps the fused kernels.  Pinned by gradients recorded from the real
reference (tests/golden/vqvae_grad.npz, tests/test_vqvae_bwd_gpu.py).
"""
from __future__ import annotations

import torch

from .. import _ffi, ops


def _pack(w, dtype):
    """Conv2d weight (O,I,kh,kw) f32 parameter -> (O,kh,kw,I) in the compute dtype."""
    return ops.repack_conv_weight(w.detach(), dtype)


def _oihw(dw):
    """packed (O,kh,kw,I) f32 gradient -> the parameter's (O,I,kh,kw)."""
    return dw.permute(0, 3, 1, 2)


def _f32(p):
    return None if p is None else p.detach().float()


class GnFn(torch.autograd.Function):
    """Normalize [+ nonlinearity] (:139-140, :164-166) on (B,H,W,C)."""

    @staticmethod
    def forward(ctx, h, gamma, beta, eps, swish):
        stats = ops.groupnorm_stats(h, eps)
        y = ops.groupnorm(h, _f32(gamma), _f32(beta), eps, swish=swish)
        ctx.save_for_backward(h, stats[0], stats[1], gamma, beta)
        ctx.swish = bool(swish)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, mean, rstd, gamma, beta = ctx.saved_tensors
        dx, dg, db = ops.groupnorm_swish_bwd(h, (mean, rstd), _f32(gamma), _f32(beta), dy.contiguous(), swish=ctx.swish)
        return dx, dg.to(gamma.dtype), db.to(beta.dtype), None, None


class Conv3x3Fn(torch.autograd.Function):
    """torch.nn.Conv2d(k = 3) sites: mode "s1" (stride 1, pad 1), "s2" (Downsample: pad (0,1,0,1) + stride 2, :151-159),
    "up" (Upsample: nearest x2 then stride 1, :171-186); optional residual (ResnetBlock's x + h, :135)."""

    @staticmethod
    def forward(ctx, h, weight, bias, residual, mode):
        wp = _pack(weight, h.dtype)
        B, H, W, _ = h.shape
        if mode == "s2":
            y = ops.conv2d_nhwc(h, wp, _f32(bias), stride=2, pad=(0, 0), out_hw=((H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1))
        else:
            y = ops.conv2d_nhwc(h, wp, _f32(bias), upsample=(mode == "up"), residual=residual)
        ctx.save_for_backward(h, wp)
        ctx.mode, ctx.has_res, ctx.has_bias = mode, residual is not None, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        h, wp = ctx.saved_tensors
        dy = dy.contiguous()
        Cin = h.shape[-1]
        need_dx = ctx.needs_input_grad[0]
        if ctx.mode == "s2":
            dx, dw, db = ops.conv3x3_s2_bwd(h, dy, wp, need_dx=need_dx)
        elif ctx.mode == "up":
            dx = ops.sumpool2(ops.conv3x3_bwd_data(dy, wp, Cin)) if need_dx else None
            dw, db = ops.conv3x3_bwd_weight(ops.upsample2(h), dy)
        else:
            dx = ops.conv3x3_bwd_data(dy, wp, Cin) if need_dx else None
            dw, db = ops.conv3x3_bwd_weight(h, dy)
        return dx, _oihw(dw), (db if ctx.has_bias else None), (dy if ctx.has_res else None), None


class Conv1x1Fn(torch.autograd.Function):
    """torch.nn.Conv2d(k = 1) sites (nin_shortcut :108-112, quant_conv / post_quant_conv :578-579): a GEMM over the pixels."""

    @staticmethod
    def forward(ctx, h, weight, bias, residual):
        B, H, W, Cin = h.shape
        Cout = weight.shape[0]
        w2 = ops.cast(weight.detach().reshape(Cout, Cin).contiguous(), h.dtype)
        y = ops.gemm(h.reshape(B * H * W, Cin), w2, bias=_f32(bias),
                     residual=None if residual is None else residual.reshape(B * H * W, Cout))
        ctx.save_for_backward(h, w2)
        ctx.has_res, ctx.has_bias = residual is not None, bias is not None
        return y.view(B, H, W, Cout)

    @staticmethod
    def backward(ctx, dy):
        h, w2 = ctx.saved_tensors
        B, H, W, Cin = h.shape
        Cout = w2.shape[0]
        d2 = dy.contiguous().reshape(B * H * W, Cout)
        dx = ops.gemm(d2, w2, b_kmajor=True).view(B, H, W, Cin) if ctx.needs_input_grad[0] else None
        dw = torch.empty(Cout, Cin, dtype=torch.float32, device=h.device)
        db = torch.empty(Cout, dtype=torch.float32, device=h.device)
        ops.wgrad(d2, h.reshape(B * H * W, Cin), dw, False, bias_out=db, bias_accumulate=False)
        return dx, dw.view(Cout, Cin, 1, 1), (db if ctx.has_bias else None), (dy if ctx.has_res else None)


class ConvInC1Fn(torch.autograd.Function):
    """Encoder.conv_in on the one-channel mel tile (:203-207): x (B,H,W) -> (B,H,W,Cout).  No gradient w.r.t. the image."""

    @staticmethod
    def forward(ctx, x, weight, bias, dtype):
        y, _ = ops.conv_in_c1(x, weight.detach().float().contiguous(), _f32(bias), dtype)
        ctx.save_for_backward(x)
        ctx.dtype, ctx.cout = dtype, weight.shape[0]
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise _ffi.MelgptError("Encoder.conv_in: no gradient w.r.t. the one-channel input image on the HIP path")
        B, H, W = x.shape
        Cout = ctx.cout
        xi = x if x.dtype == ctx.dtype else ops.cast(x.contiguous(), ctx.dtype)
        col = ops.im2col_c1(xi, ctx.dtype)                                  # (B H W, 32): taps 0 .. 8, then zeros
        d2 = dy.contiguous().reshape(B * H * W, Cout)
        dw = torch.empty(Cout, 32, dtype=torch.float32, device=x.device)
        db = torch.empty(Cout, dtype=torch.float32, device=x.device)
        ops.wgrad(d2, col, dw, False, bias_out=db, bias_accumulate=False)
        return None, dw[:, :9].reshape(Cout, 1, 3, 3), db, None


class ConvOutC1Fn(torch.autograd.Function):
    """Decoder.conv_out to ONE channel (:355-359): a (B,H,W,C) -> (B,H,W) f32.  Both gradients as GEMMs over the im2col matrix
    of the one-channel upstream gradient: da[q] = sum_t dy[q - off(t)] W[t], dW[t] = sum_q dy[q - off(t)] a[q]."""

    @staticmethod
    def forward(ctx, a, weight, bias):
        C = a.shape[-1]
        wt = ops.repack_conv_weight(weight.detach(), torch.float32).reshape(9, C)       # tap-major (9, C)
        y = ops.conv_out_c1(a, wt, _f32(bias), torch.float32)
        ctx.save_for_backward(a, wt)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, wt = ctx.saved_tensors
        B, H, W, C = a.shape
        dt = a.dtype
        dyi = dy.contiguous() if dy.dtype == dt else ops.cast(dy.contiguous(), dt)
        col = ops.im2col_c1(dyi, dt)                       # col[q][t] = dy[q + off(t)]  ->  dy[q - off(t)] = col[q][8 - t]
        wf = torch.zeros(32, C, dtype=torch.float32, device=a.device)
        wf[:9] = wt.flip(0)                                # W'[t] = W[8 - t]
        da = ops.gemm(col, ops.cast(wf, dt), b_kmajor=True).view(B, H, W, C) if ctx.needs_input_grad[0] else None
        dwf = torch.empty(32, C, dtype=torch.float32, device=a.device)
        ops.wgrad(col, a.reshape(B * H * W, C), dwf, False)                              # dwf[t'] = sum_q col[q][t'] a[q]
        dw = dwf[:9].flip(0).reshape(1, 3, 3, C).permute(0, 3, 1, 2)                     # -> (1, C, 3, 3)
        db = dy.float().sum().reshape(1)
        return da, dw, db


class AttnFn(torch.autograd.Function):
    """AttnBlock (:425-450): single-head attention over the H*W positions, scale C^-1/2, + x.  Per-image GEMMs in the
    backward (a handful of small launches per image: this path is not a measured one)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, wq, bq, wk, bk, wv, bv, wp, bp, eps):
        B, H, W, C = x.shape
        n, dt = H * W, x.dtype
        kp = (n + 7) // 8 * 8
        stats = ops.groupnorm_stats(x, eps)
        hn = ops.groupnorm(x, _f32(gamma), _f32(beta), eps, swish=False)
        wqkv = torch.empty(3 * C, C, dtype=dt, device=x.device)
        bqkv = torch.empty(3 * C, dtype=torch.float32, device=x.device)
        for i, (w_, b_) in enumerate(((wq, bq), (wk, bk), (wv, bv))):
            ops.cast(w_.detach().reshape(C, C).contiguous(), dt, out=wqkv[i * C:(i + 1) * C])
            ops.cast(b_.detach().float().contiguous(), torch.float32, out=bqkv[i * C:(i + 1) * C])
        # per image: (kp, 3C) rows, the padding rows zero - every per-image operand is a contiguous row block
        qkv = torch.zeros(B, kp, 3 * C, dtype=dt, device=x.device)
        probs = torch.empty(B, n, kp, dtype=dt, device=x.device)
        o = torch.empty(B * n, C, dtype=dt, device=x.device)
        scale = float(int(C) ** (-0.5))
        for b in range(B):
            ops.gemm(hn.view(B * n, C)[b * n:(b + 1) * n], wqkv, bias=bqkv, out=qkv[b, :n])
            q, k, v = qkv[b, :n, :C], qkv[b, :, C:2 * C], qkv[b, :, 2 * C:]
            s = torch.empty(n, kp, dtype=torch.float32, device=x.device)
            ops.gemm(q, k, out=s, alpha=scale)
            probs[b] = ops.softmax_rows(s, n, 1.0, dt, kp)
            ops.gemm(probs[b], v, b_kmajor=True, out=o[b * n:(b + 1) * n])
        wp2 = ops.cast(wp.detach().reshape(C, C).contiguous(), dt)
        y = ops.gemm(o, wp2, bias=_f32(bp), residual=x.view(B * n, C))
        ctx.save_for_backward(x, stats[0], stats[1], gamma, beta, hn, wqkv, qkv, probs, o, wp2)
        ctx.scale = scale
        return y.view(B, H, W, C)

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma, beta, hn, wqkv, qkv, probs, o, wp2 = ctx.saved_tensors
        B, H, W, C = x.shape
        n, dt, dev = H * W, x.dtype, x.device
        kp = qkv.shape[1]
        d2 = dy.contiguous().reshape(B * n, C)
        do = ops.gemm(d2, wp2, b_kmajor=True)                                           # (B n, C)
        dwp = torch.empty(C, C, dtype=torch.float32, device=dev)
        dbp = torch.empty(C, dtype=torch.float32, device=dev)
        ops.wgrad(d2, o, dwp, False, bias_out=dbp, bias_accumulate=False)
        dqkv = torch.zeros(B, kp, 3 * C, dtype=dt, device=dev)
        for b in range(B):
            q, k, v = qkv[b, :, :C], qkv[b, :, C:2 * C], qkv[b, :, 2 * C:]
            do_b = torch.zeros(kp, C, dtype=dt, device=dev)
            do_b[:n] = do[b * n:(b + 1) * n]
            dp = ops.gemm(do_b[:n], v, out_dtype=torch.float32)                        # (n, kp) = do v^T
            ds = ops.softmax_bwd_rows(probs[b], dp, n, ctx.scale)                       # (n, kp), columns >= n zero
            # rows padded to kp so that every K-major operand below is a whole number of 8-row groups; the extra rows are zero
            ds_p = torch.zeros(kp, kp, dtype=dt, device=dev)
            ds_p[:n] = ds
            pr_p = torch.zeros(kp, kp, dtype=dt, device=dev)
            pr_p[:n] = probs[b]
            ops.gemm(ds_p, k, b_kmajor=True, out=dqkv[b, :, :C])                        # dq = dS k        (rows >= n: zero)
            ops.gemm(ds_p, q, a_kmajor=True, b_kmajor=True, out=dqkv[b, :, C:2 * C])    # dk = dS^T q
            ops.gemm(pr_p, do_b, a_kmajor=True, b_kmajor=True, out=dqkv[b, :, 2 * C:])  # dv = P^T do
        dq2 = torch.empty(B * n, 3 * C, dtype=dt, device=dev)
        for b in range(B):
            dq2[b * n:(b + 1) * n] = dqkv[b, :n]
        dhn = ops.gemm(dq2, wqkv, b_kmajor=True).view(B, H, W, C)
        dwqkv = torch.empty(3 * C, C, dtype=torch.float32, device=dev)
        dbqkv = torch.empty(3 * C, dtype=torch.float32, device=dev)
        ops.wgrad(dq2, hn.view(B * n, C), dwqkv, False, bias_out=dbqkv, bias_accumulate=False)
        dxn, dg, dbeta = ops.groupnorm_swish_bwd(x, (mean, rstd), _f32(gamma), _f32(beta), dhn, swish=False)
        dx = dxn.float() + dy.float()
        dx = dx.to(dt) if ctx.needs_input_grad[0] else None
        w4 = lambda t_: t_.reshape(C, C, 1, 1)
        return (dx, dg, dbeta, w4(dwqkv[:C]), dbqkv[:C], w4(dwqkv[C:2 * C]), dbqkv[C:2 * C], w4(dwqkv[2 * C:]), dbqkv[2 * C:],
                w4(dwp), dbp, None)


# ------------------------------------------------------------------------------------------------ module-level composition
def wants_grad(module, *inputs):
    """the differentiable path is taken when autograd is recording and an input or a parameter requires a gradient"""
    if not torch.is_grad_enabled():
        return False
    if any(isinstance(x, torch.Tensor) and x.requires_grad for x in inputs):
        return True
    return any(p.requires_grad for p in module.parameters())


def gn(norm, h, swish):
    return GnFn.apply(h, norm.weight, norm.bias, norm.eps, swish)


def resnet_block(blk, h):
    """reference :114-135 (temb None, dropout p = 0)."""
    t = Conv3x3Fn.apply(gn(blk.norm1, h, True), blk.conv1.weight, blk.conv1.bias, None, "s1")
    if blk.in_channels != blk.out_channels:
        if blk.use_conv_shortcut:
            sc = Conv3x3Fn.apply(h, blk.conv_shortcut.weight, blk.conv_shortcut.bias, None, "s1")
        else:
            sc = Conv1x1Fn.apply(h, blk.nin_shortcut.weight, blk.nin_shortcut.bias, None)
    else:
        sc = h
    return Conv3x3Fn.apply(gn(blk.norm2, t, True), blk.conv2.weight, blk.conv2.bias, sc, "s1")


def attn_block(att, h):
    return AttnFn.apply(h, att.norm.weight, att.norm.bias, att.q.weight, att.q.bias, att.k.weight, att.k.bias, att.v.weight,
                        att.v.bias, att.proj_out.weight, att.proj_out.bias, att.norm.eps)


def _level(lvl, n_blocks, h):
    for i in range(n_blocks):
        h = resnet_block(lvl.block[i], h)
        if len(lvl.attn) > 0:
            h = attn_block(lvl.attn[i], h)
    return h


def encoder(enc, x, dtype):
    """reference Encoder.forward :254-282 on the differentiable path; x logical (B, in_channels, H, W) -> (B,h,w,z) NHWC."""
    if enc.in_channels == 1:
        B, _, H, W = x.shape
        xin = x.reshape(B, H, W)
        xin = xin.contiguous() if not xin.is_contiguous() else xin
        if xin.dtype not in (torch.float32, _ffi.HALF_DTYPE):
            xin = xin.float()
        h = ConvInC1Fn.apply(xin, enc.conv_in.weight, enc.conv_in.bias, dtype)
    else:
        h = Conv3x3Fn.apply(ops.to_nhwc(x, dtype), enc.conv_in.weight, enc.conv_in.bias, None, "s1")
    for i_level in range(enc.num_resolutions):
        lvl = enc.down[i_level]
        h = _level(lvl, enc.num_res_blocks, h)
        if i_level != enc.num_resolutions - 1:
            h = Conv3x3Fn.apply(h, lvl.downsample.conv.weight, lvl.downsample.conv.bias, None, "s2")
    h = resnet_block(enc.mid.block_1, h)
    h = attn_block(enc.mid.attn_1, h)
    h = resnet_block(enc.mid.block_2, h)
    return Conv3x3Fn.apply(gn(enc.norm_out, h, True), enc.conv_out.weight, enc.conv_out.bias, None, "s1")


def decoder(dec, z_nhwc):
    """reference Decoder.forward :361-392 on the differentiable path; z (B,h,w,z_channels) NHWC -> (B,H,W,out_ch) NHWC."""
    h = Conv3x3Fn.apply(z_nhwc, dec.conv_in.weight, dec.conv_in.bias, None, "s1")
    h = resnet_block(dec.mid.block_1, h)
    h = attn_block(dec.mid.attn_1, h)
    h = resnet_block(dec.mid.block_2, h)
    for i_level in reversed(range(dec.num_resolutions)):
        lvl = dec.up[i_level]
        h = _level(lvl, dec.num_res_blocks + 1, h)
        if i_level != 0:
            h = Conv3x3Fn.apply(h, lvl.upsample.conv.weight, lvl.upsample.conv.bias, None, "up")
    if dec.give_pre_end:
        return h
    h = gn(dec.norm_out, h, True)
    if dec.out_ch == 1:
        return ConvOutC1Fn.apply(h, dec.conv_out.weight, dec.conv_out.bias).unsqueeze(-1)
    return Conv3x3Fn.apply(h, dec.conv_out.weight, dec.conv_out.bias, None, "s1")


def conv1x1(conv, h):
    return Conv1x1Fn.apply(h, conv.weight, conv.bias, None)
