"""Thin tensor-level wrappers over the C ABI (include/melgpt.h).  Each function validates shapes on the
host, allocates outputs with torch (plumbing), and launches on torch's current HIP stream.  No function
here has a non-HIP fallback."""
from __future__ import annotations

import torch

from . import _ffi
from ._ffi import call, dtype_code, ptr, stream

ACT_NONE, ACT_GELU, ACT_GELU_GRAD = 0, 1, 2


def _mat(x):
    """(rows, cols) or (batch, rows, cols) tensor with unit inner stride -> (x, batch, rows, cols, ld, batch_stride)."""
    if x.dim() == 2:
        if x.stride(1) != 1:
            x = x.contiguous()
        return x, 1, x.shape[0], x.shape[1], x.stride(0), 0
    assert x.dim() == 3, x.shape
    if x.stride(2) != 1:
        x = x.contiguous()
    return x, x.shape[0], x.shape[1], x.shape[2], x.stride(1), x.stride(0)


def gemm(a, b, *, a_kmajor=False, b_kmajor=False, out=None, out_dtype=None, accumulate=False, alpha=1.0,
         bias=None, act=ACT_NONE, residual=None, pre_out=None, drop_p=0.0, seed=0, stream_id=0):
    """out[m,n] = epi(alpha * sum_k A(m,k) B(n,k)).  a: (M,K) [or (K,M) if a_kmajor]; b: (N,K) [or (K,N) if
    b_kmajor]; optional leading batch dim on both.  See melgpt_gemm in include/melgpt.h for the epilogue."""
    a, ba, ar, ac, lda, sa = _mat(a)
    b, bb, br, bc, ldb, sb = _mat(b)
    M, K = (ac, ar) if a_kmajor else (ar, ac)
    N, K2 = (bc, br) if b_kmajor else (br, bc)
    assert K == K2, (a.shape, b.shape, a_kmajor, b_kmajor)
    batch = max(ba, bb)
    assert ba in (1, batch) and bb in (1, batch)
    assert a.dtype == b.dtype, (a.dtype, b.dtype)
    dt = a.dtype
    odt = out_dtype or (out.dtype if out is not None else dt)
    assert odt in (dt, torch.float32)
    if out is None:
        shape = (batch, M, N) if (a.dim() == 3 or b.dim() == 3) else (M, N)
        out = torch.empty(shape, dtype=odt, device=a.device)
    o, bo, orow, ocol, ldc, sc = _mat(out)
    assert o is out and (orow, ocol) == (M, N) and bo in (1, batch), (out.shape, M, N)
    ldr = sr = 0
    if residual is not None:
        residual, brr, rr, rc, ldr, sr = _mat(residual)
        assert (rr, rc) == (M, N) and residual.dtype == dt
    if pre_out is not None:
        assert pre_out.shape == out.shape and pre_out.stride() == out.stride() and pre_out.dtype == out.dtype
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous()
    call("melgpt_gemm", ptr(a), int(a_kmajor), lda, sa, ptr(b), int(b_kmajor), ldb, sb, ptr(out), ldc, sc, M, N, K,
         batch, dtype_code(dt), int(odt == torch.float32 and dt != torch.float32), int(accumulate), float(alpha),
         ptr(bias), int(act), ptr(residual), ldr, sr, ptr(pre_out), float(drop_p), int(seed), int(stream_id), stream())
    return out


def conv2d_nhwc(x, wpack, bias=None, *, stride=1, pad=(1, 1), out_hw=None, upsample=False, residual=None, out=None):
    """x (B,H,W,Cin) contiguous; wpack (Cout,KH,KW,Cin) same dtype; -> (B,OH,OW,Cout).  pad = (top, left)."""
    B, H, W, Cin = x.shape
    Cout, KH, KW, Cin2 = wpack.shape
    assert Cin == Cin2 and x.is_contiguous() and wpack.is_contiguous() and x.dtype == wpack.dtype
    Hin, Win = (2 * H, 2 * W) if upsample else (H, W)
    if out_hw is None:
        out_hw = ((Hin + 2 * pad[0] - KH) // stride + 1, (Win + 2 * pad[1] - KW) // stride + 1)
    OH, OW = out_hw
    if out is None:
        out = torch.empty(B, OH, OW, Cout, dtype=x.dtype, device=x.device)
    assert out.shape == (B, OH, OW, Cout) and out.is_contiguous()
    if residual is not None:
        assert residual.shape == out.shape and residual.is_contiguous() and residual.dtype == x.dtype
    call("melgpt_conv2d_nhwc", ptr(x), B, H, W, Cin, ptr(wpack), Cout, KH, KW, stride, pad[0], pad[1], OH, OW,
         int(upsample), ptr(bias), ptr(residual), ptr(out), dtype_code(x.dtype), stream())
    return out
