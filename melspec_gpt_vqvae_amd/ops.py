"""Thin tensor-level wrappers over the C ABI (include/melgpt.h).  Each function validates shapes on the
host, allocates outputs with torch (plumbing), and launches on torch's current HIP stream.  No function
here has a non-HIP fallback."""
from __future__ import annotations

import os

import torch

from . import _ffi
from ._ffi import call, dtype_code, ptr, stream

ACT_NONE, ACT_GELU, ACT_GELU_GRAD, ACT_GELU_DACT, ACT_MUL = 0, 1, 2, 3, 4


class KernelTimer:
    """Live per-launch timing of the MFMA GEMM / conv kernel with HIP events recorded on the launch stream
    (used by bench.py for the roofline figure; off by default - two event records per launch)."""

    def __init__(self):
        self.records = []  # (start_event, end_event, flops, tag)
        self.aux = []      # HBM-bound launches timed beside the family: (start_event, end_event, bytes, flops, tag)

    def aux_by_tag(self):
        """{tag: (launches, total_ms, algorithmic bytes, flops)} of the launches recorded with `_timed_aux`."""
        agg = {}
        for s, e, nbytes, f, tag in self.aux:
            a = agg.setdefault(tag, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += s.elapsed_time(e)
            a[2] += nbytes
            a[3] += f
        return {t: tuple(v) for t, v in agg.items()}

    def summary(self):
        """total_ms = the time during which AT LEAST ONE recorded launch was running: the union of the launches' intervals on
        the device clock (launches of a Block's weight gradients run on a second stream beside the input-gradient chain -
        transformer/minGPT.py _wgrad - so the plain sum of durations would count shared time twice); serial_ms = that sum."""
        flops = sum(f for _, _, f, _ in self.records)
        if not self.records:
            return dict(launches=0, total_ms=0.0, serial_ms=0.0, flops=flops)
        base = self.records[0][0]
        iv = sorted((base.elapsed_time(s), base.elapsed_time(e)) for s, e, _, _ in self.records)
        union, cur_a, cur_b = 0.0, iv[0][0], iv[0][1]
        for a, b in iv[1:]:
            if a > cur_b:
                union += cur_b - cur_a
                cur_a, cur_b = a, b
            else:
                cur_b = max(cur_b, b)
        union += cur_b - cur_a
        return dict(launches=len(self.records), total_ms=union, serial_ms=sum(b - a for a, b in iv), flops=flops)

    def by_tag(self):
        agg = {}
        for s, e, f, tag in self.records:
            ms = s.elapsed_time(e)
            a = agg.setdefault(tag, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += ms
            a[2] += f
        return sorted(((t, n, ms, fl) for t, (n, ms, fl) in agg.items()), key=lambda r: -r[2])


TIMER = None  # set to a KernelTimer() to record


class _timed:
    def __init__(self, flops, tag):
        self.flops, self.tag = flops, tag

    def __enter__(self):
        if TIMER is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def cancel(self):
        """the call was refused before any launch (MELGPT_ERR_UNSUPPORTED): its FLOPs belong to the fallback's record"""
        self.flops = None

    def __exit__(self, *a):
        if TIMER is not None and self.flops is not None:
            self.e.record()
            TIMER.records.append((self.s, self.e, self.flops, self.tag))
        return False


class _timed_aux:
    """the same two event records around a launch that is NOT of the GEMM family (attention): its algorithmic HBM
    bytes and FLOPs go to KernelTimer.aux, never into the family's sums"""

    def __init__(self, nbytes, flops, tag):
        self.nbytes, self.flops, self.tag = nbytes, flops, tag

    def __enter__(self):
        if TIMER is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        if TIMER is not None and a[0] is None:
            self.e.record()
            TIMER.aux.append((self.s, self.e, self.nbytes, self.flops, self.tag))
        return False


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
    with _timed(2.0 * M * N * K * batch, f"gemm {'T' if a_kmajor else 'N'}{'N' if b_kmajor else 'T'} {M}x{N}x{K}"
                + (f" b{batch}" if batch > 1 else "") + ({ACT_GELU: " gelu", ACT_GELU_GRAD: " dgelu", ACT_GELU_DACT: " gelu+dact", ACT_MUL: " mul"}.get(act, ""))
                + (" drop" if drop_p > 0 else "")
                + (" res" if residual is not None and act not in (ACT_GELU_GRAD, ACT_MUL) else "")):
        call("melgpt_gemm", ptr(a), int(a_kmajor), lda, sa, ptr(b), int(b_kmajor), ldb, sb, ptr(out), ldc, sc, M, N, K,
             batch, dtype_code(dt), int(odt == torch.float32 and dt != torch.float32), int(accumulate), float(alpha),
             ptr(bias), int(act), ptr(residual), ldr, sr, ptr(pre_out), float(drop_p), int(seed), int(stream_id),
             stream())
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
    with _timed(2.0 * B * OH * OW * Cout * KH * KW * Cin, f"conv{KH}x{KW} s{stride} {H}x{W} {Cin}->{Cout}"):
        call("melgpt_conv2d_nhwc", ptr(x), B, H, W, Cin, ptr(wpack), Cout, KH, KW, stride, pad[0], pad[1], OH, OW,
             int(upsample), ptr(bias), ptr(residual), ptr(out), dtype_code(x.dtype), stream())
    return out


# --------------------------------------------------------------------------------- workspace
_WS = {}


def workspace(nfloats, device):
    """A cached f32 scratch buffer (grown on demand).  All kernels run on the current stream in program
    order, so one buffer per device is enough for the two-stage reductions."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream())
    buf = _WS.get(key)
    if buf is None or buf.numel() < nfloats:
        buf = torch.empty(max(int(nfloats), 1 << 20), dtype=torch.float32, device=device)
        _WS[key] = buf
    return buf


# --------------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x, gamma, beta, eps=1e-5, want_stats=True):
    """x (M,C) contiguous -> (y, mean, rstd)."""
    M, C = x.shape
    assert x.is_contiguous() and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    y = torch.empty_like(x)
    mean = torch.empty(M, dtype=torch.float32, device=x.device) if want_stats else None
    rstd = torch.empty(M, dtype=torch.float32, device=x.device) if want_stats else None
    call("melgpt_layernorm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), M, C, float(eps),
         dtype_code(x.dtype), stream())
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, *, add_in=None, dgamma=None, dbeta=None, accumulate=False, mask=None):
    """returns dx (= add_in + LN'(dy)); writes dgamma/dbeta (f32, (+)= if accumulate) when given.
    mask = (drop_p, seed, stream_id): returns (dx, dropout_apply(dx, drop_p, seed, stream_id)) from the same pass - the
    mask replay of the branch that consumes dx next."""
    M, C = x.shape
    assert dy.is_contiguous() and x.is_contiguous() and dy.dtype == x.dtype
    dx = torch.empty_like(x)
    ws = None
    if dgamma is not None:
        nw = _ffi.lib().melgpt_layernorm_bwd_nwaves(M)
        ws = workspace(nw * 2 * C, x.device)
    if mask is None:
        call("melgpt_layernorm_bwd", ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(add_in), ptr(dx),
             ptr(dgamma), ptr(dbeta), int(accumulate), ptr(ws), M, C, dtype_code(x.dtype), stream())
        return dx
    drop_p, seed, stream_id = mask
    dxm = torch.empty_like(x)
    call("melgpt_layernorm_bwd_masked", ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(add_in), ptr(dx),
         ptr(dgamma), ptr(dbeta), int(accumulate), ptr(ws), M, C, ptr(dxm), float(drop_p), int(seed), int(stream_id),
         dtype_code(x.dtype), stream())
    return dx, dxm


def colsum(a, out, accumulate=False):
    """out[n] (+)= sum_m a[m,n]; a (M,N) with unit inner stride, out f32 (N,)."""
    M, N = a.shape
    assert a.stride(1) == 1 and out.dtype == torch.float32 and out.numel() == N
    ws = workspace(_ffi.lib().melgpt_colsum_rows() * N, a.device)
    call("melgpt_colsum", ptr(a), M, N, a.stride(0), ptr(out), int(accumulate), ptr(ws), dtype_code(a.dtype), stream())
    return out


# --------------------------------------------------------------------------------- embedding stem
def embed_fwd(idx, tok_emb, pos_emb, *, pre_idx=None, pre_table=None, pre_vals=None, n_pre=0, dtype=torch.float32,
              drop_p=0.0, seed=0, stream_id=0):
    """idx (B,Tt) int64; tok_emb (V,C) f32; pos_emb (>=Tt+n_pre, C) f32 -> (B, Tt+n_pre, C) in `dtype`."""
    B, Tt = idx.shape
    V, C = tok_emb.shape
    assert tok_emb.dtype == torch.float32 and tok_emb.is_contiguous() and pos_emb.is_contiguous()
    assert pos_emb.shape[-1] == C and pos_emb.shape[-2] >= Tt + n_pre
    if idx.stride(-1) != 1:
        idx = idx.contiguous()
    out = torch.empty(B, Tt + n_pre, C, dtype=dtype, device=tok_emb.device)
    if pre_idx is not None:
        pre_idx = pre_idx.reshape(-1).contiguous()
        assert pre_idx.numel() == B * n_pre and pre_table.is_contiguous() and pre_table.dtype == torch.float32
    if pre_vals is not None:
        assert pre_vals.shape == (B, n_pre, C) and pre_vals.dtype == torch.float32 and pre_vals.is_contiguous()
    call("melgpt_embed_fwd", ptr(idx) if Tt > 0 else None, ptr(tok_emb), ptr(pos_emb), ptr(pre_idx), ptr(pre_table),
         ptr(pre_vals), n_pre, B, Tt, idx.stride(0) if Tt > 0 else 0, C, V, ptr(out), dtype_code(dtype), float(drop_p), int(seed), int(stream_id), stream())
    return out


def embed_bwd(dx, idx, *, tok_grad=None, pos_grad=None, pre_idx=None, pre_table_grad=None, pre_vals_grad=None,
              n_pre=0, accumulate=False, drop_p=0.0, seed=0, stream_id=0):
    B, Ttot, C = dx.shape
    Tt = Ttot - n_pre
    assert dx.is_contiguous()
    V = tok_grad.shape[0] if tok_grad is not None else 1
    n_rows = pre_table_grad.shape[0] if pre_table_grad is not None else 0
    if pre_idx is not None:
        pre_idx = pre_idx.reshape(-1).contiguous()
    if idx.stride(-1) != 1:
        idx = idx.contiguous()
    call("melgpt_embed_bwd", ptr(dx), ptr(idx) if Tt > 0 else None, idx.stride(0) if Tt > 0 else 0, ptr(pre_idx), n_pre,
         B, Tt, C, V,
         n_rows, ptr(tok_grad), ptr(pos_grad), ptr(pre_table_grad), ptr(pre_vals_grad), int(accumulate),
         dtype_code(dx.dtype), float(drop_p), int(seed), int(stream_id), stream())


def _split_count(rows, target=32):
    """largest divisor of `rows` that is <= target (split-K factor for the embedding-gradient GEMM)."""
    for d in range(min(target, rows), 0, -1):
        if rows % d == 0:
            return d
    return 1


_CUS = None


def _persistent_workgroups():
    """workgroups a persistent-kernel launch gets right now: the device's CUs minus melgpt_set_reserved_cus()"""
    global _CUS
    if _CUS is None:
        _CUS = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    left = _CUS - _ffi.lib().melgpt_get_reserved_cus()
    return left if left >= 8 else _CUS


def _wgrad_split(M, N, K, dtype):
    """split-K factor for the weight-gradient GEMM (a divisor of the M reduction rows).
    bf16, big outputs: the persistent 256x256 kernel runs one workgroup per CU, so the factor is chosen to fill whole
    rounds of that many tiles (cost = rounds x rows per batch, plus the pass that sums the f32 partials);
    otherwise: ~3 workgroups of the 128x128 kernel per CU."""
    t256 = ((N + 255) // 256) * ((K + 255) // 256)
    if dtype == _ffi.HALF_DTYPE and t256 >= 12:
        best, best_cost = 1, None
        wgs = _persistent_workgroups()
        for ns in range(1, 33):
            if M % ns or (M // ns) % 8 or (ns > 1 and M // ns < 512):
                continue
            rounds = -(-(t256 * ns) // wgs)
            if t256 * ns < 192:          # the dispatcher keeps such launches on the 128x128 kernel
                continue
            cost = rounds * (M // ns) * 33e-9 + (ns > 1) * (ns + 1) * N * K * 4 / 3e12
            if best_cost is None or cost < best_cost:
                best, best_cost = ns, cost
        if best_cost is not None:
            return best
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    want = max(1, min(16, 768 // max(tiles, 1)))
    return _split_count(M, want) if want > 1 else 1


def wgrad(dy, x, out, accumulate, bias_out=None, bias_accumulate=None):
    """out (N,K) f32 (+)= dy^T (N x M) @ x (M x K): the weight gradient of y = x W^T.  The reduction runs over
    M = B*T rows (tens of thousands) while the output has only a few dozen to a few hundred tiles, so the rows are
    split into batches (_wgrad_split); partial products are summed in fixed order (deterministic).
    bias_out (N,) f32: also (+)= the column sums of dy - the bias gradient - taken in the same launch when the shape
    runs on the persistent kernel (melgpt_wgrad_rowsum), by melgpt_colsum otherwise."""
    M, N = dy.shape
    K = x.shape[1]
    if bias_accumulate is None:
        bias_accumulate = accumulate
    ns = _wgrad_split(M, N, K, dy.dtype)
    if ns == 1 or dy.stride(1) != 1 or x.stride(1) != 1 or (M // ns) < 512:
        if bias_out is not None:
            colsum(dy, bias_out, accumulate=bias_accumulate)
        return gemm(dy, x, a_kmajor=True, b_kmajor=True, out=out, accumulate=accumulate)
    rows = M // ns
    if bias_out is not None and dy.dtype == _ffi.HALF_DTYPE and dy.dtype == x.dtype:
        L = _ffi.lib()
        nrs = L.melgpt_wgrad_rowsum_rows(K, ns)
        part = torch.empty(ns, N, K, dtype=torch.float32, device=dy.device)
        rpart = torch.empty(nrs, N, dtype=torch.float32, device=dy.device)
        with _timed(2.0 * M * N * K, f"gemm TN {N}x{K}x{rows} b{ns} +rowsum") as tm:
            code = L.melgpt_wgrad_rowsum(ptr(dy), dy.stride(0), rows * dy.stride(0), ptr(x), x.stride(0), rows * x.stride(0),
                                         ptr(part), K, N * K, N, K, rows, ns, dtype_code(dy.dtype), ptr(rpart), N, stream())
            if code == _ffi.ERR_UNSUPPORTED:
                tm.cancel()
        if code != _ffi.ERR_UNSUPPORTED:
            _ffi.check(code, "melgpt_wgrad_rowsum")
            call("melgpt_reduce_rows_pair", ptr(part), ns, N * K, N * K, ptr(out), int(accumulate),
                 ptr(rpart), nrs, N, N, ptr(bias_out), int(bias_accumulate), stream())
            return out
    if bias_out is not None:
        colsum(dy, bias_out, accumulate=bias_accumulate)
    a3 = torch.as_strided(dy, (ns, rows, N), (rows * dy.stride(0), dy.stride(0), 1), dy.storage_offset())
    b3 = torch.as_strided(x, (ns, rows, K), (rows * x.stride(0), x.stride(0), 1), x.storage_offset())
    part = gemm(a3, b3, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)     # (ns, N, K)
    call("melgpt_reduce_rows", ptr(part), ns, N * K, N * K, ptr(out), int(accumulate), 1.0, stream())
    return out


def embed_table_grad(dx, idx, tok_grad, *, n_pre=0, accumulate=False, drop_p=0.0, seed=0, stream_id=0):
    """d tok_emb (V,C) (+)= OneHot^T (V x M) @ keep(dX) (M x C) on the MFMA GEMM: K = B*T rows are split into
    batches (deterministic split-K: partial tables are summed in fixed order by melgpt_reduce_rows)."""
    B, Ttot, C = dx.shape
    V = tok_grad.shape[0]
    Tt = Ttot - n_pre
    if idx.stride(-1) != 1:
        idx = idx.contiguous()
    M = B * Ttot
    dxm = dropout_apply(dx, drop_p, seed, stream_id) if drop_p > 0 else dx
    oh = torch.empty(M, V, dtype=dx.dtype, device=dx.device)
    call("melgpt_onehot_rows", ptr(idx), idx.stride(0), B, Tt, n_pre, V, ptr(oh), dtype_code(dx.dtype), stream())
    ns = _split_count(M)
    part = gemm(oh.view(ns, M // ns, V), dxm.view(ns, M // ns, C), a_kmajor=True, b_kmajor=True,
                out_dtype=torch.float32)                                            # (ns, V, C)
    call("melgpt_reduce_rows", ptr(part), ns, V * C, V * C, ptr(tok_grad), int(accumulate), 1.0, stream())
    return dxm


# --------------------------------------------------------------------------------- cross entropy
def cross_entropy_fwd(logits, target):
    """logits (M,V) f32 (unit inner stride), target (M,) int64 -> (loss_rows (M,), lse (M,))."""
    M, V = logits.shape
    assert logits.dtype == torch.float32 and logits.stride(1) == 1
    target = target.reshape(-1).contiguous()
    loss = torch.empty(M, dtype=torch.float32, device=logits.device)
    lse = torch.empty(M, dtype=torch.float32, device=logits.device)
    call("melgpt_cross_entropy_fwd", ptr(logits), logits.stride(0), ptr(target), M, V, ptr(loss), ptr(lse), stream())
    return loss, lse


def cross_entropy_bwd(logits, target, lse, *, g_rows=None, g_group=1, g_scalar=None, g_scale=1.0, dtype=torch.float32):
    M, V = logits.shape
    target = target.reshape(-1).contiguous()
    d = torch.empty(M, V, dtype=dtype, device=logits.device)
    call("melgpt_cross_entropy_bwd", ptr(logits), logits.stride(0), ptr(target), ptr(lse), ptr(g_rows), int(g_group),
         ptr(g_scalar), float(g_scale), M, V, ptr(d), V, dtype_code(dtype), stream())
    return d


def group_sum(x, n, scale=1.0):
    """x (groups*n,) f32 -> (groups,) sums of consecutive runs of n."""
    x = x.reshape(-1)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() % n == 0
    out = torch.empty(x.numel() // n, dtype=torch.float32, device=x.device)
    call("melgpt_group_sum_f32", ptr(x), x.numel() // n, n, float(scale), ptr(out), stream())
    return out


def sample_logits(logits, temperature=1.0, top_k=None, sample=False, seed=0, step=0, want_probs=False):
    """logits (rows, V) f32 -> next-token ids (rows, 1) int64 (one step of the reference's sampling loops)."""
    assert logits.dim() == 2 and logits.dtype == torch.float32 and logits.stride(1) == 1
    rows, V = logits.shape
    out = torch.empty(rows, 1, dtype=torch.int64, device=logits.device)
    probs = torch.empty(rows, V, dtype=torch.float32, device=logits.device) if want_probs else None
    call("melgpt_sample_logits", ptr(logits), logits.stride(0), rows, V, float(temperature), int(top_k or 0), int(sample),
         int(seed), int(step), ptr(out), ptr(probs), stream())
    return (out, probs) if want_probs else out


def sum_f32(x, scale=1.0, out=None, accumulate=False):
    x = x.reshape(-1)
    assert x.dtype == torch.float32 and x.is_contiguous()
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=x.device)
    call("melgpt_sum_f32", ptr(x), x.numel(), float(scale), ptr(out), int(accumulate), stream())
    return out


def dropout_apply(x, drop_p, seed, stream_id):
    assert x.is_contiguous()
    y = torch.empty_like(x)
    call("melgpt_dropout_apply", ptr(x), ptr(y), x.numel(), float(drop_p), int(seed), int(stream_id),
         dtype_code(x.dtype), stream())
    return y


def dropout_apply_colsum(x, drop_p, seed, stream_id, out, accumulate=False):
    """-> y = dropout_apply(x, ...), and out[n] (+)= sum_m y[m, n] from the same pass (x (M,N) contiguous)."""
    M, N = x.shape
    assert x.is_contiguous() and out.dtype == torch.float32 and out.numel() == N
    y = torch.empty_like(x)
    ws = workspace(_ffi.lib().melgpt_colsum_rows() * N, x.device)
    call("melgpt_dropout_apply_colsum", ptr(x), ptr(y), M, N, float(drop_p), int(seed), int(stream_id), ptr(out),
         int(accumulate), ptr(ws), dtype_code(x.dtype), stream())
    return y


def zero_(x):
    """x.zero_() through the library (x contiguous; an empty tensor is left alone)."""
    assert x.is_contiguous()
    if x.numel():
        call("melgpt_zero_bytes", ptr(x), x.numel() * x.element_size(), stream())
    return x


def cast(x, dtype, out=None):
    assert x.is_contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=dtype, device=x.device)
    assert out.is_contiguous() and out.numel() == x.numel()
    call("melgpt_cast", ptr(x), dtype_code(x.dtype), ptr(out), dtype_code(out.dtype), x.numel(), stream())
    return out


def adamw(param, grad, exp_avg, exp_avg_sq, *, lr, betas, eps, weight_decay, step, param_bf16=None, grad_scale=1.0):
    n = param.numel()
    for t_ in (param, grad, exp_avg, exp_avg_sq):
        assert t_.dtype == torch.float32 and t_.is_contiguous() and t_.numel() == n
    call("melgpt_adamw", ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), ptr(param_bf16), n, float(lr),
         float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step), float(grad_scale), stream())


# --------------------------------------------------------------------------------- attention
def attn_fwd(q, k, v, n_head, *, B, T, n_unmasked=0, drop_p=0.0, seed=0, stream_id=0, want_att=False):
    """q,k,v: (B*T, C) views with a common row stride (e.g. column blocks of the packed QKV matrix)."""
    M, C = q.shape
    assert M == B * T and C == n_head * 64 and q.stride(1) == 1
    assert q.stride(0) == k.stride(0) == v.stride(0) and q.dtype == k.dtype == v.dtype
    out = torch.empty(M, C, dtype=q.dtype, device=q.device)
    lse = torch.empty(B, n_head, T, dtype=torch.float32, device=q.device)
    att = torch.empty(B, n_head, T, T, dtype=torch.float32, device=q.device) if want_att else None
    es = q.element_size()
    # algorithmic bytes: q, k, v read once, y written once, one f32 log-sum-exp per (head, row) [+ the map when asked for];
    # FLOPs on the full T x T square (4 T^2 C per sequence and layer, no causal discount)
    with _timed_aux(4.0 * M * C * es + 4.0 * B * n_head * T * (1 + (T if want_att else 0)), 4.0 * B * T * T * C,
                    f"attn fwd {B}x{n_head}x{T}" + (" full" if n_unmasked >= T else "") + (" +att" if want_att else "")):
        call("melgpt_attn_fwd", ptr(q), ptr(k), ptr(v), q.stride(0), ptr(out), C, ptr(lse), ptr(att), B, n_head, T, 64,
             int(n_unmasked), float(drop_p), int(seed), int(stream_id), dtype_code(q.dtype), stream())
    return out, lse, att


def attn_decode(qkv, kcache, vcache, n_head, pos, att_row=None, pos_dev=None, out=None):
    """one KV-cached attention step: qkv (B, 3C) [key|query|value] of the new token; caches (B, Tmax, C), appended at
    `pos` (or at *pos_dev, a device int32, when given); -> (B, C)."""
    B, C3 = qkv.shape
    C = C3 // 3
    assert qkv.stride(1) == 1 and kcache.shape == vcache.shape and kcache.shape[0] == B and kcache.shape[2] == C
    assert kcache.is_contiguous() and vcache.is_contiguous() and kcache.dtype == qkv.dtype
    if out is None:
        out = torch.empty(B, C, dtype=qkv.dtype, device=qkv.device)
    call("melgpt_attn_decode", ptr(qkv), qkv.stride(0), ptr(kcache), ptr(vcache), B, n_head, 64, kcache.shape[1], int(pos),
         ptr(pos_dev), ptr(out), ptr(att_row), dtype_code(qkv.dtype), stream())
    return out


SKINNY_LN_MAX_ROWS = int(os.environ.get("MELGPT_SKINNY_LN_MAX_ROWS", "64"))  # lab switch (0: never fuse)
LDS_LINEAR_MIN_ROWS = int(os.environ.get("MELGPT_LDS_LINEAR_MIN_ROWS", "5"))  # lab switch (999: never).  Default from the
# end-to-end chain (profiles/r03_decode_lab.md): sampling 265 tokens at 16 sequences 238.8 -> 210.5 ms, at 8 sequences
# 216.0 -> 202.4 ms with the LDS-resident linear instead of the register-pipelined one (whose per-kernel timing at <= 16
# rows had looked equal in isolation)


_LN_FOLD = {}   # (weight, bias, gamma, beta versions) -> (W' bf16, c1, c2): melgpt_ln_fold_prepare, rebuilt when any changes


def _ln_folded(w, bias, gamma, beta):
    from .flat import SHADOW_EPOCH, tensor_version

    key = (w.data_ptr(), tuple(w.shape), w.stride(0))
    ver = (tensor_version(w), SHADOW_EPOCH[0], tensor_version(bias), tensor_version(gamma), tensor_version(beta))
    hit = _LN_FOLD.get(key)
    if hit is None or hit[0] != ver:
        N, K = w.shape
        wf = torch.empty(N, K, dtype=w.dtype, device=w.device)
        c1 = torch.empty(N, dtype=torch.float32, device=w.device)
        c2 = torch.empty(N, dtype=torch.float32, device=w.device)
        call("melgpt_ln_fold_prepare", ptr(w), w.stride(0), ptr(bias), ptr(gamma.detach()), ptr(beta.detach()), N, K,
             dtype_code(w.dtype), ptr(wf), ptr(c1), ptr(c2), stream())
        if len(_LN_FOLD) > 256:
            _LN_FOLD.clear()
        hit = _LN_FOLD[key] = (ver, wf, c1, c2)
    return hit[1], hit[2], hit[3]


def linear_rows(x, w, *, bias=None, act=ACT_NONE, residual=None, out_dtype=None, ln=None):
    """y (M,N) = epi(x (M,K) @ w (N,K)^T + bias) (+ residual) for a handful of rows (decode steps): the weight-streaming
    kernel melgpt_gemv_rows instead of the tiled MFMA GEMM.  ln = (gamma, beta, eps): the rows are LayerNorm-ed on the
    way in (the block's pre-LN folded into its qkv / fc1 layer)."""
    M, K = x.shape
    N, K2 = w.shape
    assert K == K2 and x.dtype == w.dtype and x.stride(1) == 1 and w.stride(1) == 1
    odt = out_dtype or x.dtype
    assert odt in (x.dtype, torch.float32)
    skinny = x.dtype == _ffi.HALF_DTYPE and 4 < M <= 128 and N % 16 == 0 and K % 128 == 0
    lds_ok = x.dtype == _ffi.HALF_DTYPE and LDS_LINEAR_MIN_ROWS <= M <= 128 and N % 16 == 0
    if lds_ok and K % 1024 == 0 and K <= 8192 and (ln is None or K == 1024):
        # 17 .. 128 rows: x through LDS by LDS-DMA, weights to registers, everything requested up front (melgpt_linear_lds)
        y = torch.empty(M, N, dtype=odt, device=x.device)
        if residual is not None:
            assert residual.shape == (M, N) and residual.dtype == x.dtype and residual.stride(1) == 1
        c1 = c2 = None
        eps = 0.0
        if ln is not None:   # the pre-LN folded into the weight (prepared once per version of weight / LayerNorm / bias)
            w, c1, c2 = _ln_folded(w, bias, ln[0], ln[1])
            eps = ln[2]
        nws = int(_ffi.lib().melgpt_linear_lds_workspace(M, N, K))
        ws = workspace((nws + 3) // 4, x.device) if nws else None
        call("melgpt_linear_lds", ptr(x), x.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(residual),
             residual.stride(0) if residual is not None else 0, ptr(y), N, M, N, K, int(act), dtype_code(x.dtype),
             int(odt == torch.float32), ptr(c1), ptr(c2), float(eps), ptr(ws), stream())
        return y
    skinny_ln = skinny and ln is not None and M <= SKINNY_LN_MAX_ROWS and K in (512, 1024)
    if ln is not None and M > 4 and not skinny_ln:
        # the weight-streaming kernels re-derive the row statistics in every workgroup: that pays for a few rows (one
        # launch less per LayerNorm) but not for many - normalise once, separately
        x, ln = layernorm_fwd(x.contiguous(), ln[0], ln[1], ln[2], want_stats=False)[0], None
    if M > 32 and not skinny:  # enough rows for the MFMA tiles to pay (measured: 128 rows 5.0 vs 5.8 ms per 24-layer step)
        return gemm(x, w, bias=bias, act=act, residual=residual, out_dtype=odt)
    y = torch.empty(M, N, dtype=odt, device=x.device)
    if residual is not None:
        assert residual.shape == (M, N) and residual.dtype == x.dtype and residual.stride(1) == 1
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous()
    g = b = None
    eps = 0.0
    if ln is not None:
        g, b, eps = ln
        assert g.dtype == torch.float32 and b.dtype == torch.float32 and g.numel() == K and b.numel() == K
    if skinny:  # 5 .. 128 bf16 rows: N / 16 workgroups stream the weights once as MFMA operands
        call("melgpt_linear_skinny", ptr(x), x.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(residual),
             residual.stride(0) if residual is not None else 0, ptr(y), N, M, N, K, int(act), dtype_code(x.dtype),
             int(odt == torch.float32), ptr(g), ptr(b), float(eps), stream())
        return y
    call("melgpt_gemv_rows", ptr(x), x.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(residual),
         residual.stride(0) if residual is not None else 0, ptr(y), N, M, N, K, int(act), dtype_code(x.dtype),
         int(odt == torch.float32), ptr(g), ptr(b), float(eps), stream())
    return y


def embed_decode(idx, tok_emb, pos_emb, pos_dev, dtype):
    """x (B, C) = tok_emb[idx[:, 0]] + pos_emb[*pos_dev]; pos_emb (block_size, C) f32, pos_dev device int32."""
    B = idx.shape[0]
    V, C = tok_emb.shape
    assert idx.dtype == torch.int64 and idx.is_contiguous() and pos_emb.is_contiguous() and pos_dev.dtype == torch.int32
    out = torch.empty(B, C, dtype=dtype, device=tok_emb.device)
    call("melgpt_embed_decode", ptr(idx), ptr(tok_emb), ptr(pos_emb), ptr(pos_dev), B, C, V, ptr(out), dtype_code(dtype),
         stream())
    return out


def incr_i32(counter):
    assert counter.dtype == torch.int32 and counter.numel() == 1
    call("melgpt_incr_i32", ptr(counter), stream())


def sample_logits_dev(logits, out, pos_dev, step_offset, temperature=1.0, top_k=None, sample=False, seed=0, seq=None):
    """sample_logits with the step number (*pos_dev + step_offset) read on the device; writes `out` (rows,1) int64 in
    place and, when given, seq[:, *pos_dev]."""
    assert logits.dim() == 2 and logits.dtype == torch.float32 and logits.stride(1) == 1
    rows, V = logits.shape
    assert out.dtype == torch.int64 and out.numel() == rows and out.is_contiguous()
    if seq is not None:
        assert seq.dtype == torch.int64 and seq.stride(1) == 1 and seq.shape[0] == rows
    call("melgpt_sample_logits_dev", ptr(logits), logits.stride(0), rows, V, float(temperature), int(top_k or 0),
         int(sample), int(seed), ptr(pos_dev), int(step_offset), ptr(out), ptr(seq), seq.stride(0) if seq is not None else 0,
         stream())
    return out


def attn_bwd(q, k, v, out, dout, lse, n_head, *, B, T, dqkv=None, n_unmasked=0, drop_p=0.0, seed=0, stream_id=0):
    """-> (dq, dk, dv) as column blocks [q | k | v] of one (B*T, 3C) buffer unless dqkv views are given."""
    M, C = q.shape
    assert out.is_contiguous() and dout.is_contiguous() and dout.dtype == q.dtype
    if dqkv is None:
        buf = torch.empty(M, 3 * C, dtype=q.dtype, device=q.device)
        dqkv = (buf[:, :C], buf[:, C:2 * C], buf[:, 2 * C:])
    dq, dk, dv = dqkv
    assert dq.stride(0) == dk.stride(0) == dv.stride(0)
    delta = torch.empty(B, n_head, T, dtype=torch.float32, device=q.device)
    es = q.element_size()
    # algorithmic bytes: q, k, v, y, dy read once, dq, dk, dv written once, lse read + delta written and read;
    # FLOPs by BASELINE.md 3's convention forward : backward = 1 : 2 (the kernel evaluates five products - S, dP, dV, dK, dQ)
    with _timed_aux(8.0 * M * C * es + 12.0 * B * n_head * T, 8.0 * B * T * T * C,
                    f"attn bwd {B}x{n_head}x{T}" + (" full" if n_unmasked >= T else "")):
        call("melgpt_attn_bwd", ptr(q), ptr(k), ptr(v), q.stride(0), ptr(out), ptr(dout), C, ptr(lse), ptr(delta), ptr(dq),
             ptr(dk), ptr(dv), dq.stride(0), B, n_head, T, 64, int(n_unmasked), float(drop_p), int(seed), int(stream_id),
             dtype_code(q.dtype), stream())
    return dq, dk, dv


# --------------------------------------------------------------------------------- VQ-VAE pieces (NHWC)
def groupnorm_stats(x, eps=1e-6):
    """x (B,H,W,C) contiguous -> (mean, rstd) of GroupNorm(32), each (B*32,) f32."""
    B, H, W, C = x.shape
    assert x.is_contiguous()
    L = _ffi.lib()
    nch = L.melgpt_groupnorm_nchunks(H * W)
    ws = workspace(B * nch * 64 + 2 * B * 32, x.device)
    mean = torch.empty(B * 32, dtype=torch.float32, device=x.device)
    rstd = torch.empty(B * 32, dtype=torch.float32, device=x.device)
    call("melgpt_groupnorm_stats", ptr(x), B, H * W, C, float(eps), ptr(mean), ptr(rstd), ptr(ws), dtype_code(x.dtype),
         stream())
    return mean, rstd


def fused_conv_supported(Cin, dtype, occupancy=1):
    """kernel limit (occupancy=1) or the profitable regime (occupancy=2: two workgroups per CU, measured on MI355X:
    128-channel bf16 layers 605 TFLOP/s incl. the norm; 256-channel layers at one workgroup per CU only 290-350)."""
    es = 4 if dtype == torch.float32 else 2
    lds = 180 * Cin * es + 32768 + Cin * 8
    return Cin % (32 if es == 4 else 64) == 0 and Cin * es >= 256 and lds * occupancy <= 160 * 1024


def conv3x3_gn(x, stats, gamma, beta, wpack, bias, *, swish=True, residual=None):
    """ResnetBlock's norm -> swish -> conv3x3 in one launch.  x (B,H,W,Cin) raw; stats = (mean, rstd) or None."""
    B, H, W, Cin = x.shape
    Cout = wpack.shape[0]
    assert x.is_contiguous() and wpack.is_contiguous() and wpack.shape[1:] == (3, 3, Cin) and wpack.dtype == x.dtype
    out = torch.empty(B, H, W, Cout, dtype=x.dtype, device=x.device)
    mean, rstd = stats if stats is not None else (None, None)
    with _timed(2.0 * B * H * W * Cout * 9 * Cin, f"conv3x3+gn {H}x{W} {Cin}->{Cout}"):
        call("melgpt_conv3x3_gn_nhwc", ptr(x), B, H, W, Cin, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), int(swish),
             ptr(wpack), Cout, ptr(bias), ptr(residual), ptr(out), dtype_code(x.dtype), stream())
    return out


def conv3x3_gn_with_out_stats(x, stats, gamma, beta, wpack, bias, out_eps, *, swish=True, residual=None):
    """conv3x3_gn that also returns the GroupNorm(32) statistics (mean, rstd) of its OUTPUT (residual included),
    accumulated in the conv's epilogue (the next norm then needs no pass over the tensor).  Returns None when this
    shape does not run on the persistent fused kernel (bf16, Cout 128) - nothing has been launched then."""
    B, H, W, Cin = x.shape
    Cout = wpack.shape[0]
    if x.dtype != _ffi.HALF_DTYPE or Cout != 128:
        return None
    assert x.is_contiguous() and wpack.is_contiguous() and wpack.shape[1:] == (3, 3, Cin) and wpack.dtype == x.dtype
    L = _ffi.lib()
    ws = workspace(L.melgpt_conv3x3_gn_stats_workspace(B, H, W), x.device)
    out = torch.empty(B, H, W, Cout, dtype=x.dtype, device=x.device)
    omean = torch.empty(B * 32, dtype=torch.float32, device=x.device)
    orstd = torch.empty(B * 32, dtype=torch.float32, device=x.device)
    mean, rstd = stats if stats is not None else (None, None)
    with _timed(2.0 * B * H * W * Cout * 9 * Cin, f"conv3x3+gn {H}x{W} {Cin}->{Cout}") as tm:
        code = L.melgpt_conv3x3_gn_nhwc_stats(ptr(x), B, H, W, Cin, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), int(swish),
                                              ptr(wpack), Cout, ptr(bias), ptr(residual), ptr(out), dtype_code(x.dtype),
                                              float(out_eps), ptr(omean), ptr(orstd), ptr(ws), stream())
        if code == _ffi.ERR_UNSUPPORTED:
            tm.cancel()
    if code == _ffi.ERR_UNSUPPORTED:
        return None
    _ffi.check(code, "melgpt_conv3x3_gn_nhwc_stats")
    return out, (omean, orstd)


def groupnorm(x, gamma, beta, eps=1e-6, swish=True):
    """x (B,H,W,C) contiguous -> GroupNorm(32) [+ swish]."""
    B, H, W, C = x.shape
    assert x.is_contiguous()
    y = torch.empty_like(x)
    # small images (the 10 x 106 and 5 x 53 levels): statistics + normalisation in one launch, the tensor read once
    code = _ffi.lib().melgpt_groupnorm_fused(ptr(x), ptr(gamma), ptr(beta), ptr(y), B, H * W, C, float(eps), int(swish),
                                             None, None, dtype_code(x.dtype), stream())
    if code != _ffi.ERR_UNSUPPORTED:
        _ffi.check(code, "melgpt_groupnorm_fused")
        return y
    mean, rstd = groupnorm_stats(x, eps)
    call("melgpt_groupnorm_apply", ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(y), B, H * W, C, int(swish),
         dtype_code(x.dtype), stream())
    return y


# ----- backward of the VQ-VAE's building blocks (csrc/vqvae_bwd.hip): C-ABI exports with parity tests, NOT wired into
# LitVQVAE's autograd (no scored configuration trains the VQ-VAE; .backward() through the module is refused loudly)
def conv3x3_bwd_data(dy, wpack, Cin):
    """dy (B,H,W,Cout), wpack (Cout,3,3,Cin) in dy's dtype -> dx (B,H,W,Cin): input gradient of conv3x3 (stride 1, pad 1)."""
    B, H, W, Cout = dy.shape
    assert dy.is_contiguous() and wpack.is_contiguous() and wpack.shape == (Cout, 3, 3, Cin) and wpack.dtype == dy.dtype
    nbytes = int(_ffi.lib().melgpt_conv3x3_bwd_workspace(B, H, W, Cin, Cout, dtype_code(dy.dtype)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
    dx = torch.empty(B, H, W, Cin, dtype=dy.dtype, device=dy.device)
    call("melgpt_conv3x3_bwd_data", ptr(dy), ptr(wpack), ptr(dx), B, H, W, Cin, Cout, ptr(ws), dtype_code(dy.dtype), stream())
    return dx


def conv3x3_bwd_weight(x, dy, want_bias=True):
    """x (B,H,W,Cin), dy (B,H,W,Cout) -> (dw (Cout,3,3,Cin) f32 in the packed layout, dbias (Cout,) f32 or None)."""
    B, H, W, Cin = x.shape
    Cout = dy.shape[3]
    assert x.is_contiguous() and dy.is_contiguous() and dy.shape[:3] == x.shape[:3] and dy.dtype == x.dtype
    nbytes = int(_ffi.lib().melgpt_conv3x3_bwd_workspace(B, H, W, Cin, Cout, dtype_code(x.dtype)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    dw = torch.empty(Cout, 3, 3, Cin, dtype=torch.float32, device=x.device)
    db = torch.empty(Cout, dtype=torch.float32, device=x.device) if want_bias else None
    call("melgpt_conv3x3_bwd_weight", ptr(x), ptr(dy), ptr(dw), ptr(db), B, H, W, Cin, Cout, ptr(ws), dtype_code(x.dtype), stream())
    return dw, db


def conv3x3_s2_bwd(x, dy, wpack, need_dx=True):
    """Downsample's convolution (pad (0,1,0,1) + conv3x3 stride 2): x (B,H,W,Cin), dy (B,OH,OW,Cout), wpack (Cout,3,3,Cin) ->
    (dx or None, dw (Cout,3,3,Cin) f32, dbias (Cout,) f32)."""
    B, H, W, Cin = x.shape
    Cout = dy.shape[3]
    assert x.is_contiguous() and dy.is_contiguous() and wpack.is_contiguous() and dy.dtype == x.dtype == wpack.dtype
    assert dy.shape[1:3] == ((H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1)
    ws = torch.empty(int(_ffi.lib().melgpt_conv3x3_s2_bwd_workspace(B, H, W, Cin, Cout, dtype_code(x.dtype))), dtype=torch.uint8, device=x.device)
    dx = None
    if need_dx:
        dx = torch.empty_like(x)
        call("melgpt_conv3x3_s2_bwd_data", ptr(dy), ptr(wpack), ptr(dx), B, H, W, Cin, Cout, ptr(ws), dtype_code(x.dtype), stream())
    dw = torch.empty(Cout, 3, 3, Cin, dtype=torch.float32, device=x.device)
    db = torch.empty(Cout, dtype=torch.float32, device=x.device)
    call("melgpt_conv3x3_s2_bwd_weight", ptr(x), ptr(dy), ptr(dw), ptr(db), B, H, W, Cin, Cout, ptr(ws), dtype_code(x.dtype), stream())
    return dx, dw, db


def upsample2(x):
    """x (B,H,W,C) -> (B,2H,2W,C), nearest."""
    B, H, W, C = x.shape
    assert x.is_contiguous()
    y = torch.empty(B, 2 * H, 2 * W, C, dtype=x.dtype, device=x.device)
    call("melgpt_upsample2_nhwc", ptr(x), ptr(y), B, H, W, C, dtype_code(x.dtype), stream())
    return y


def sumpool2(x):
    """x (B,2H,2W,C) -> (B,H,W,C): sums of the 2 x 2 blocks (the adjoint of upsample2)."""
    B, H2, W2, C = x.shape
    assert x.is_contiguous() and H2 % 2 == 0 and W2 % 2 == 0
    y = torch.empty(B, H2 // 2, W2 // 2, C, dtype=x.dtype, device=x.device)
    call("melgpt_sumpool2_nhwc", ptr(x), ptr(y), B, H2 // 2, W2 // 2, C, dtype_code(x.dtype), stream())
    return y


def softmax_bwd_rows(probs, dprobs, n, scale):
    """probs (..., rows, ld) in the compute dtype (columns >= n zero), dprobs (..., rows, ld) f32 -> dscores like probs."""
    assert probs.is_contiguous() and dprobs.is_contiguous() and dprobs.dtype == torch.float32 and probs.shape == dprobs.shape
    ld = probs.shape[-1]
    rows = probs.numel() // ld
    ds = torch.empty_like(probs)
    call("melgpt_softmax_bwd_rows", ptr(probs), ld, ptr(dprobs), ld, int(n), rows, float(scale), ptr(ds), ld, dtype_code(probs.dtype), stream())
    return ds


def im2col_c1(img, dtype):
    """img (B,H,W) in `dtype` -> (B*H*W, 32): the nine taps of a 3x3 / pad 1 convolution of a one-channel image, zero-padded to 32."""
    B, H, W = img.shape
    assert img.is_contiguous() and img.dtype == dtype
    out = torch.empty(B * H * W, 32, dtype=dtype, device=img.device)
    call("melgpt_im2col_c1", ptr(img), ptr(out), B, H, W, dtype_code(dtype), stream())
    return out


def groupnorm_swish_bwd(x, stats, gamma, beta, dy, swish=True):
    """gradient of swish?(GroupNorm(32)(x) * gamma + beta): x, dy (B,H,W,C); stats = (mean, rstd) of the forward
    (groupnorm_stats) -> (dx, dgamma, dbeta)."""
    B, H, W, C = x.shape
    mean, rstd = stats
    assert x.is_contiguous() and dy.is_contiguous() and dy.shape == x.shape and dy.dtype == x.dtype
    nbytes = int(_ffi.lib().melgpt_groupnorm_swish_bwd_workspace(B, H * W, C))
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    dg = torch.empty(C, dtype=torch.float32, device=x.device)
    db = torch.empty(C, dtype=torch.float32, device=x.device)
    call("melgpt_groupnorm_swish_bwd", ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(dy), ptr(dx), ptr(dg), ptr(db), B, H * W,
         C, int(swish), ptr(ws), dtype_code(x.dtype), stream())
    return dx, dg, db


def conv_in_c1(x, w, bias, dtype, stats_eps=None):
    """x (B,H,W) f32/bf16 single-channel image; w (Cout,1,3,3) f32 -> (B,H,W,Cout) in dtype.  stats_eps (Cout = 128):
    also the GroupNorm(32) statistics of the output from the same pass -> (y, (mean, rstd)); else (y, None)."""
    B, H, W = x.shape
    Cout = w.shape[0]
    assert x.is_contiguous() and w.is_contiguous() and w.dtype == torch.float32
    y = torch.empty(B, H, W, Cout, dtype=dtype, device=x.device)
    if stats_eps is not None and Cout == 128:
        ws = workspace(_ffi.lib().melgpt_conv_in_c1_stats_workspace(B, H, W), x.device)
        mean = torch.empty(B * 32, dtype=torch.float32, device=x.device)
        rstd = torch.empty(B * 32, dtype=torch.float32, device=x.device)
        code = _ffi.lib().melgpt_conv_in_c1_stats(ptr(x), dtype_code(x.dtype), ptr(w), ptr(bias), ptr(y), dtype_code(dtype), B, H,
                                                  W, Cout, float(stats_eps), ptr(mean), ptr(rstd), ptr(ws), stream())
        if code != _ffi.ERR_UNSUPPORTED:   # (an image too wide for the kernel's LDS window: the plain stem, statistics by the caller)
            _ffi.check(code, "melgpt_conv_in_c1_stats")
            return y, (mean, rstd)
    call("melgpt_conv_in_c1", ptr(x), dtype_code(x.dtype), ptr(w), ptr(bias), ptr(y), dtype_code(dtype), B, H, W, Cout,
         stream())
    return y, None


def conv_out_c1(x, w_tap_major, bias, out_dtype=torch.float32):
    B, H, W, C = x.shape
    assert x.is_contiguous() and w_tap_major.dtype == torch.float32 and w_tap_major.numel() == 9 * C
    y = torch.empty(B, H, W, dtype=out_dtype, device=x.device)
    call("melgpt_conv_out_c1", ptr(x), dtype_code(x.dtype), ptr(w_tap_major), ptr(bias), ptr(y), dtype_code(out_dtype),
         B, H, W, C, stream())
    return y


def softmax_rows(scores, n, scale, out_dtype, ld_out):
    """scores (..., rows, ld) f32 -> probs (..., rows, ld_out) with columns >= n zeroed."""
    assert scores.dtype == torch.float32 and scores.is_contiguous()
    rows = scores.numel() // scores.shape[-1]
    probs = torch.empty(scores.shape[:-1] + (ld_out,), dtype=out_dtype, device=scores.device)
    call("melgpt_softmax_rows", ptr(scores), scores.shape[-1], n, rows, float(scale), ptr(probs), ld_out,
         dtype_code(out_dtype), stream())
    return probs


def repack_conv_weight(w, dtype):
    """(O,I,KH,KW) f32 -> (O,KH,KW,I) in dtype."""
    O, I, KH, KW = w.shape
    w = w.detach()
    assert w.dtype == torch.float32 and w.is_contiguous()
    out = torch.empty(O, KH, KW, I, dtype=dtype, device=w.device)
    call("melgpt_repack_conv_weight", ptr(w), ptr(out), dtype_code(dtype), O, I, KH, KW, stream())
    return out


def to_nhwc(x, dtype):
    """logical (B,C,H,W) tensor (any strides) -> contiguous (B,H,W,C) in dtype; free when it already is."""
    B, C, H, W = x.shape
    v = x.permute(0, 2, 3, 1)
    if v.is_contiguous() and x.dtype == dtype:
        return v
    if not x.is_contiguous():
        if v.is_contiguous():  # channels-last, wrong dtype
            return cast(v, dtype)
        x = x.contiguous()
    y = torch.empty(B, H, W, C, dtype=dtype, device=x.device)
    call("melgpt_permute_nchw_nhwc", ptr(x), dtype_code(x.dtype), ptr(y), dtype_code(dtype), B, C, H * W, 1, stream())
    return y


def to_nchw_contiguous(x_nhwc, dtype):
    B, H, W, C = x_nhwc.shape
    y = torch.empty(B, C, H, W, dtype=dtype, device=x_nhwc.device)
    call("melgpt_permute_nchw_nhwc", ptr(x_nhwc), dtype_code(x_nhwc.dtype), ptr(y), dtype_code(dtype), B, C, H * W, 0,
         stream())
    return y


def codes_permute(codes, H, W, reverse=False):
    """(B,H,W) or (B,H*W) int64 codes -> (B, H*W) in the other ordering (time-major <-> row-major)."""
    B = codes.shape[0]
    src = codes.reshape(B, H * W).contiguous()
    assert src.dtype == torch.int64
    out = torch.empty_like(src)
    call("melgpt_codes_permute", ptr(src), ptr(out), B, H, W, int(reverse), stream())
    return out


# --------------------------------------------------------------------------------- MelGAN generator glue
def pad1d_act(x, pad, *, reflect=True, slope=1.0):
    """x (B, L, C) contiguous -> (B, L + 2 pad, C): reflection (or zero) padding along L with LeakyReLU(slope) applied
    on the way (slope = 1: none)."""
    B, L, C = x.shape
    assert x.is_contiguous()
    y = torch.empty(B, L + 2 * pad, C, dtype=x.dtype, device=x.device)
    call("melgpt_pad1d_act", ptr(x), ptr(y), B, L, C, int(pad), int(reflect), float(slope), dtype_code(x.dtype), stream())
    return y


def conv1d_nlc(x, wcat, bias, taps, *, dilation=1, pad_l=0, reflect=False, in_slope=0.0, residual=None, out=None,
               accumulate=False, out_slope=0.0):
    """x (B, L, Cin) contiguous, wcat (Cout, taps * Cin) tap-major -> (B, L, Cout): one implicit-GEMM launch
    (melgpt_conv1d_nlc): y[l] = bias + sum_t W_t f(x[l - pad_l + t dilation]) (+ residual) (+ out when accumulate),
    f = LeakyReLU(in_slope) (0: none), out-of-range positions reflected or zero; out_slope != 0: LeakyReLU on bias + sum.  `out` may be a row-strided view
    (B, L, Cout) of a larger tensor - a transposed convolution's output phase - as long as its rows are uniformly spaced."""
    B, L, Cin = x.shape
    Cout = wcat.shape[0]
    assert x.is_contiguous() and wcat.is_contiguous() and wcat.shape[1] == taps * Cin and wcat.dtype == x.dtype
    if out is None:
        out = torch.empty(B, L, Cout, dtype=x.dtype, device=x.device)
    assert out.shape == (B, L, Cout) and out.stride(2) == 1 and out.stride(0) == L * out.stride(1) and out.dtype == x.dtype
    ldr = 0
    if residual is not None:
        assert residual.shape == (B, L, Cout) and residual.stride(2) == 1 and residual.stride(0) == L * residual.stride(1)
        ldr = residual.stride(1)
    call("melgpt_conv1d_nlc", ptr(x), B, L, Cin, ptr(wcat), Cout, int(taps), int(dilation), int(pad_l), int(bool(reflect)),
         float(in_slope), ptr(bias), ptr(residual), ldr, int(bool(accumulate)), float(out_slope), ptr(out), out.stride(1),
         dtype_code(x.dtype), stream())
    return out


def resblock_narrow(x, wfrag, b3, b1s, dilation, slope):
    """x (B, L, C) 16-bit, C in {32, 64} with L % 16 == 0 or C == 128 with L % 64 == 0 -> y (B, L, C): one MelGAN ResnetBlock in one pass
    (melgpt_resblock_narrow; wfrag / b3 / b1s as vocoder.modules packs them)."""
    B, L, C = x.shape
    assert x.is_contiguous() and wfrag.is_contiguous() and b3.dtype == torch.float32 and b1s.dtype == torch.float32
    y = torch.empty_like(x)
    call("melgpt_resblock_narrow", ptr(x), ptr(y), ptr(wfrag), ptr(b3), ptr(b1s), B, L, C, int(dilation), float(slope),
         dtype_code(x.dtype), stream())
    return y


def conv1d_out1_fused(h, w, bias, K, slope, tanh=True):
    """h (B, L, C) 16-bit, w (K*C,) f32 tap-major -> (B, L) f32 = [tanh](conv to one channel of reflect_pad(leaky(h)))."""
    B, L, C = h.shape
    assert h.is_contiguous() and w.dtype == torch.float32 and w.numel() == K * C and w.is_contiguous()
    y = torch.empty(B, L, dtype=torch.float32, device=h.device)
    call("melgpt_conv1d_out1_fused", ptr(h), ptr(w), ptr(bias), ptr(y), B, L, C, K, float(slope), int(tanh),
         dtype_code(h.dtype), stream())
    return y


def conv1d_out1(xp, w, bias, L, K, tanh=True):
    """xp (B, L + K - 1, C) padded activation, w (K*C,) f32 -> (B, L) f32 = [tanh](conv to one channel)."""
    B, Lp, C = xp.shape
    assert Lp == L + K - 1 and xp.is_contiguous() and w.dtype == torch.float32 and w.numel() == K * C and w.is_contiguous()
    y = torch.empty(B, L, dtype=torch.float32, device=xp.device)
    call("melgpt_conv1d_out1", ptr(xp), ptr(w), ptr(bias), ptr(y), B, L, C, K, int(tanh), dtype_code(xp.dtype), stream())
    return y
