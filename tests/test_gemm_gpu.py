"""GPU parity of the MFMA GEMM / implicit-GEMM convolution (csrc/gemm.hip) against plain fp32 PyTorch on CPU.
Tolerances: f32 lane 1e-5 relative-to-max (exact f32 FMA chains, only the summation order differs);
bf16 lane: inputs are pre-rounded to bf16 on both sides, products exact, f32 accumulate -> 1e-5 before the
output rounding, 2^-8 relative after it."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import synth
from util import rel_err, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DT = {"f32": torch.float32, "bf16": torch.bfloat16}


def _rand(seed, shape, dt, scale=1.0):
    x = t(synth.normal(seed, shape) * np.float32(scale)).to(dt)
    return x.to(DEV), x.float()


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (530, 128, 256), (265, 384, 96), (77, 20, 40), (1, 4, 8),
                                    (300, 1024, 512)])
def test_nt_plain(dt, M, N, K):
    from melspec_gpt_vqvae_amd import ops

    a, ac = _rand(1, (M, K), DT[dt])
    b, bc = _rand(2, (N, K), DT[dt])
    out = ops.gemm(a, b, out_dtype=torch.float32)
    assert rel_err(out.cpu().numpy(), (ac @ bc.t()).numpy()) < 1e-5


# GPT-VAE XL widths (config_GPT_VAE_vggsound.py:43-58: C = 1472 = 23 * 64, not a multiple of 128 / 256):
# qkv N = 4416, fc1 N = 5888, fc2 K = 5888, the encoder's head N = 2944, V = 1024
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("form,M,N,K", [("nt", 530, 1472, 1472), ("nt", 530, 4416, 1472), ("nt", 530, 5888, 1472),
                                        ("nt", 530, 1472, 5888), ("nt", 265, 2944, 1472), ("nt", 530, 1024, 1472),
                                        ("nn", 530, 1472, 5888), ("nn", 530, 5888, 1472), ("nn", 530, 1472, 2944),
                                        ("tn", 1472, 5888, 530), ("tn", 2944, 1472, 530), ("tn", 5888, 1472, 530)])
def test_xl_width_shapes(dt, form, M, N, K):
    from melspec_gpt_vqvae_amd import ops

    if form == "nt":
        a, ac = _rand(40, (M, K), DT[dt], 0.5)
        b, bc = _rand(41, (N, K), DT[dt], 0.5)
        out, ref = ops.gemm(a, b, out_dtype=torch.float32), ac @ bc.t()
    elif form == "nn":
        a, ac = _rand(42, (M, K), DT[dt], 0.5)
        b, bc = _rand(43, (K, N), DT[dt], 0.5)
        out, ref = ops.gemm(a, b, b_kmajor=True, out_dtype=torch.float32), ac @ bc
    else:   # weight gradient dW (M x N) = dY^T X over K = tokens, through the split-K path the step uses
        a, ac = _rand(44, (K, M), DT[dt], 0.5)
        b, bc = _rand(45, (K, N), DT[dt], 0.5)
        out = torch.empty(M, N, device=DEV)
        ops.wgrad(a, b, out, False)
        ref = ac.t() @ bc
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 1e-5


def test_xl_width_persistent_kernel_at_step_size():
    """the XL shapes at the token count of one rank's step slice (M = 64 * 265), bf16, on the persistent kernel"""
    from melspec_gpt_vqvae_amd import ops

    M = 64 * 265
    torch.manual_seed(9)
    a = (torch.randn(M, 1472) * 0.5).to(torch.bfloat16)
    for N in (4416, 5888, 1472):
        b = (torch.randn(N, 1472) * 0.2).to(torch.bfloat16)
        out = ops.gemm(a.to(DEV), b.to(DEV), out_dtype=torch.float32)
        assert rel_err(out.cpu().numpy(), (a.float() @ b.float().t()).numpy()) < 1e-5
    a2 = (torch.randn(M, 5888) * 0.5).to(torch.bfloat16)
    b2 = (torch.randn(1472, 5888) * 0.2).to(torch.bfloat16)
    out = ops.gemm(a2.to(DEV), b2.to(DEV), out_dtype=torch.float32)
    assert rel_err(out.cpu().numpy(), (a2.float() @ b2.float().t()).numpy()) < 1e-5
    dy = (torch.randn(M, 5888) * 0.1).to(torch.bfloat16).to(DEV)
    gw = torch.empty(5888, 1472, device=DEV)
    ops.wgrad(dy, a.to(DEV), gw, False)
    assert rel_err(gw.cpu().numpy(), (dy.float().cpu().t() @ a.float()).numpy()) < 1e-5


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("form", ["nn", "tn", "tt"])
def test_transposed_operands(dt, form):
    """nn: dX = dY W (B k-major);  tn: dW = dY^T X (both k-major);  tt: A k-major, B row."""
    from melspec_gpt_vqvae_amd import ops

    M, N, K = 200, 136, 328
    if form == "nn":
        a, ac = _rand(3, (M, K), DT[dt])
        b, bc = _rand(4, (K, N), DT[dt])
        out = ops.gemm(a, b, b_kmajor=True, out_dtype=torch.float32)
        ref = ac @ bc
    elif form == "tn":
        a, ac = _rand(5, (K, M), DT[dt])
        b, bc = _rand(6, (K, N), DT[dt])
        out = ops.gemm(a, b, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
        ref = ac.t() @ bc
    else:
        a, ac = _rand(7, (K, M), DT[dt])
        b, bc = _rand(8, (N, K), DT[dt])
        out = ops.gemm(a, b, a_kmajor=True, out_dtype=torch.float32)
        ref = ac.t() @ bc.t()
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 1e-5


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_asymmetric_identity_layout_check(dt):
    """A = I with an asymmetric B catches a transposed C write (guide §3)."""
    from melspec_gpt_vqvae_amd import ops

    n = 128
    a = torch.eye(n, dtype=DT[dt], device=DEV)
    bc = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 61) - 30 + torch.arange(n)[:, None] * 0.5
    out = ops.gemm(a, bc.to(DT[dt]).to(DEV), out_dtype=torch.float32)
    assert torch.equal(out.cpu(), bc.to(DT[dt]).float().t().contiguous())


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_epilogues(dt):
    from melspec_gpt_vqvae_amd import ops

    M, N, K = 265, 256, 128
    a, ac = _rand(10, (M, K), DT[dt], 0.5)
    b, bc = _rand(11, (N, K), DT[dt], 0.5)
    bias = t(synth.normal(12, (N,)), DEV)
    r, rc = _rand(13, (M, N), DT[dt])
    lin = ac @ bc.t() * 0.37 + bias.cpu()
    tol = 1e-5 if dt == "f32" else 6e-3
    # bias + GELU, with pre-activation copy
    pre = torch.empty(M, N, dtype=DT[dt], device=DEV)
    out = ops.gemm(a, b, alpha=0.37, bias=bias, act=ops.ACT_GELU, pre_out=pre)
    assert rel_err(pre.float().cpu().numpy(), lin.numpy()) < tol
    assert rel_err(out.float().cpu().numpy(), F.gelu(lin).numpy()) < tol
    # bias + residual
    out = ops.gemm(a, b, alpha=0.37, bias=bias, residual=r)
    assert rel_err(out.float().cpu().numpy(), (lin + rc).numpy()) < tol
    # gelu-grad epilogue: v * gelu'(R)
    x = rc.clone().requires_grad_(True)
    F.gelu(x).sum().backward()
    out = ops.gemm(a, b, alpha=0.37, act=ops.ACT_GELU_GRAD, residual=r, out_dtype=torch.float32)
    assert rel_err(out.cpu().numpy(), ((ac @ bc.t() * 0.37) * x.grad).numpy()) < 2e-5
    # accumulate into f32
    acc = t(synth.normal(14, (M, N)), DEV)
    acc0 = acc.cpu().clone()
    ops.gemm(a, b, out=acc, accumulate=True)
    assert rel_err(acc.cpu().numpy(), (acc0 + ac @ bc.t()).numpy()) < 1e-5
    # strided views (q inside a packed qkv buffer) and batched
    big, bigc = _rand(15, (M, 3 * K), DT[dt], 0.5)
    out = ops.gemm(big[:, K:2 * K], b, out_dtype=torch.float32)
    assert rel_err(out.cpu().numpy(), (bigc[:, K:2 * K] @ bc.t()).numpy()) < 1e-5
    a3, a3c = _rand(16, (3, 70, K), DT[dt], 0.5)
    b3, b3c = _rand(17, (3, 52, K), DT[dt], 0.5)
    out = ops.gemm(a3, b3, out_dtype=torch.float32)
    assert rel_err(out.cpu().numpy(), torch.bmm(a3c, b3c.transpose(1, 2)).numpy()) < 1e-5


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(265, 256, 128), (16640, 1536, 256)])
def test_gelu_with_saved_derivative_and_mul_epilogues(dt, M, N, K):
    """forward of Linear -> GELU writing gelu(pre) and gelu'(pre) (MELGPT_ACT_GELU_DACT), backward multiplying by the
    saved derivative (MELGPT_ACT_MUL) == the two-erf form (ACT_GELU + ACT_GELU_GRAD on the stored pre-activation):
    bit-identical on the f32 lane, within bf16 rounding of the stored tensor on the bf16 lane; small (128-tile kernel)
    and multi-round persistent-kernel sizes."""
    from melspec_gpt_vqvae_amd import ops

    a, ac = _rand(50, (M, K), DT[dt], 0.5)
    b, bc = _rand(51, (N, K), DT[dt], 0.5)
    bias = t(synth.normal(52, (N,)), DEV)
    pre_ref = ac @ bc.t() + bias.cpu()
    x = pre_ref.clone().requires_grad_(True)
    F.gelu(x).sum().backward()
    der_ref = x.grad
    dact = torch.empty(M, N, dtype=DT[dt], device=DEV)
    act = ops.gemm(a, b, bias=bias, act=ops.ACT_GELU_DACT, pre_out=dact)
    pre = torch.empty(M, N, dtype=DT[dt], device=DEV)
    act_old = ops.gemm(a, b, bias=bias, act=ops.ACT_GELU, pre_out=pre)
    assert torch.equal(act, act_old)
    tol = 1e-5 if dt == "f32" else 2 ** -8
    assert rel_err(act.float().cpu().numpy(), F.gelu(pre_ref).numpy()) < tol
    assert rel_err(dact.float().cpu().numpy(), der_ref.numpy()) < tol
    d, dc = _rand(53, (M, 192), DT[dt], 0.5)
    w2, w2c = _rand(54, (192, N), DT[dt], 0.2)                      # (out, in) weight used K-major in the dgrad product
    g_new = ops.gemm(d, w2, b_kmajor=True, act=ops.ACT_MUL, residual=dact)
    g_old = ops.gemm(d, w2, b_kmajor=True, act=ops.ACT_GELU_GRAD, residual=pre)
    if dt == "f32":
        assert torch.equal(g_new, g_old)
    ref = (dc @ w2c) * der_ref
    assert rel_err(g_new.float().cpu().numpy(), ref.numpy()) < (1e-5 if dt == "f32" else 2 ** -7)
    # dropout + residual after the activation still apply to C (not to the saved derivative)
    r, rc = _rand(55, (M, N), DT[dt])
    dact2 = torch.empty_like(dact)
    y = ops.gemm(a, b, bias=bias, act=ops.ACT_GELU_DACT, pre_out=dact2, residual=r)
    assert torch.equal(dact2, dact)
    assert rel_err(y.float().cpu().numpy(), (F.gelu(pre_ref) + rc).numpy()) < (1e-5 if dt == "f32" else 2 ** -7)


def test_dropout_epilogue_statistics_and_replay():
    from melspec_gpt_vqvae_amd import ops

    M, N, K = 512, 512, 64
    a, ac = _rand(20, (M, K), torch.float32, 0.5)
    b, bc = _rand(21, (N, K), torch.float32, 0.5)
    ref = (ac @ bc.t()).numpy()
    o1 = ops.gemm(a, b, drop_p=0.5, seed=1234, stream_id=7).cpu().numpy()
    o2 = ops.gemm(a, b, drop_p=0.5, seed=1234, stream_id=7).cpu().numpy()
    o3 = ops.gemm(a, b, drop_p=0.5, seed=1234, stream_id=8).cpu().numpy()
    assert np.array_equal(o1, o2), "counter-based mask must replay exactly"
    keep = o1 != 0
    assert abs(keep.mean() - 0.5) < 0.01
    assert rel_err(o1[keep], 2.0 * ref[keep]) < 1e-5
    assert (keep != (o3 != 0)).mean() > 0.4, "different stream id -> independent mask"
    o4 = ops.gemm(a, b, drop_p=0.3, seed=99, stream_id=1).cpu().numpy()
    assert abs((o4 != 0).mean() - 0.7) < 0.01
    # rows / columns are not correlated
    k2 = (o4 != 0).astype(np.float64)
    assert abs(np.corrcoef(k2[:, 0], k2[:, 4])[0, 1]) < 0.2 and abs(np.corrcoef(k2[0], k2[1])[0, 1]) < 0.2


def _conv_ref(xc, wc, bias, stride, pad, ups, res):
    x = xc.permute(0, 3, 1, 2)
    if ups:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    if stride == 2:
        x = F.pad(x, (0, 1, 0, 1))
        y = F.conv2d(x, wc, bias, stride=2, padding=0)
    else:
        y = F.conv2d(x, wc, bias, stride=1, padding=pad)
    y = y.permute(0, 2, 3, 1)
    return y + res if res is not None else y


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("case", ["3x3", "3x3_res", "down", "up", "1x1"])
def test_conv_implicit_gemm(dt, case):
    from melspec_gpt_vqvae_amd import ops

    B, H, W, Cin, Cout = 2, 10, 53, 128, 64
    k = 1 if case == "1x1" else 3
    x, xc = _rand(30, (B, H, W, Cin), DT[dt])
    w, wc = _rand(31, (Cout, Cin, k, k), DT[dt], 0.05)
    bias = t(synth.normal(32, (Cout,)), DEV)
    wpack = w.permute(0, 2, 3, 1).contiguous()
    stride, pad, ups, res, resc = 1, (k // 2, k // 2), False, None, None
    if case == "down":
        stride, pad = 2, (0, 0)
    if case == "up":
        ups = True
    oh = {"down": (5, 26), "up": (20, 106)}.get(case, (H, W))
    if case == "3x3_res":
        res, resc = _rand(33, (B, H, W, Cout), DT[dt])
    y = ops.conv2d_nhwc(x, wpack, bias, stride=stride, pad=pad, out_hw=oh, upsample=ups, residual=res)
    ref = _conv_ref(xc, wc, bias.cpu(), stride, k // 2, ups, resc)
    assert y.shape == ref.shape
    assert rel_err(y.float().cpu().numpy(), ref.numpy()) < (1e-5 if dt == "f32" else 6e-3)


# ------------------------------------------------------------------ persistent wide-tile kernel at multi-round sizes
# (csrc/gemm256.hip is picked automatically for bf16 problems with >= 192 tiles of 256x256: every workgroup then
# walks several tiles, so the ring that keeps prefetching across tile boundaries, the epilogue staged through a ring
# slot and both tile heights are exercised; reference = fp32 matmul of the bf16-rounded operands on CPU)
@pytest.mark.parametrize("form,M,N,K", [("nt", 20000, 1280, 384), ("nt", 33920, 1024, 264), ("nn", 24576, 1024, 320),
                                        ("nt", 9000, 4096, 328)])
def test_wide_kernel_many_tiles_per_workgroup(form, M, N, K):
    from melspec_gpt_vqvae_amd import ops

    torch.manual_seed(M + N + K)
    a = (torch.randn(M, K) * 0.5).to(torch.bfloat16)
    if form == "nt":
        b = (torch.randn(N, K) * 0.5).to(torch.bfloat16)
        ref = a.float() @ b.float().t()
        out = ops.gemm(a.to(DEV), b.to(DEV), out_dtype=torch.float32)
    else:
        b = (torch.randn(K, N) * 0.5).to(torch.bfloat16)
        ref = a.float() @ b.float()
        out = ops.gemm(a.to(DEV), b.to(DEV), b_kmajor=True, out_dtype=torch.float32)
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 1e-5


def test_wide_kernel_fused_epilogues_and_split_k_at_size():
    from melspec_gpt_vqvae_amd import ops

    M, N, K = 16640, 1536, 256          # 65 x 6 = 390 tiles of 256 rows (512 of 192): two rounds
    torch.manual_seed(3)
    a = (torch.randn(M, K) * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K) * 0.2).to(torch.bfloat16)
    bias = torch.randn(N) * 0.1
    res = torch.randn(M, N).to(torch.bfloat16)
    pre_ref = a.float() @ b.float().t() + bias
    ad, bd = a.to(DEV), b.to(DEV)
    # bias + exact GELU + second output (fc1 of the MLP)
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    act = ops.gemm(ad, bd, bias=bias.to(DEV), act=ops.ACT_GELU, pre_out=pre)
    assert rel_err(pre.float().cpu().numpy(), pre_ref.numpy()) < 2 ** -8
    assert rel_err(act.float().cpu().numpy(), F.gelu(pre_ref).numpy()) < 2 ** -8
    # bias + residual (plain mode) and bias + dropout + residual (full mode; mask replayed by dropout_apply)
    y = ops.gemm(ad, bd, bias=bias.to(DEV), residual=res.to(DEV))
    assert rel_err(y.float().cpu().numpy(), (pre_ref + res.float()).numpy()) < 2 ** -8
    yd = ops.gemm(ad, bd, bias=bias.to(DEV), residual=res.to(DEV), drop_p=0.25, seed=77, stream_id=5)
    ones = torch.ones(M, N, dtype=torch.bfloat16, device=DEV)
    mask = ops.dropout_apply(ones, 0.25, 77, 5).float().cpu()            # 0 or 1/(1-p)
    assert abs(float((mask > 0).float().mean()) - 0.75) < 5e-3
    assert rel_err(yd.float().cpu().numpy(), (pre_ref * mask + res.float()).numpy()) < 2 ** -7
    # GELU' epilogue of the backward (R = pre-activation)
    g = ops.gemm(ad, bd, act=ops.ACT_GELU_GRAD, residual=pre)
    x = pre.float().cpu()
    gprime = 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
    assert rel_err(g.float().cpu().numpy(), ((a.float() @ b.float().t()) * gprime).numpy()) < 2 ** -7
    # split-K weight gradient (K-major operands, f32 partials summed in fixed order), bit-reproducible
    dy = (torch.randn(M, 1024) * 0.1).to(torch.bfloat16).to(DEV)
    xw = (torch.randn(M, 768) * 0.5).to(torch.bfloat16).to(DEV)
    gw = torch.empty(1024, 768, device=DEV)
    ops.wgrad(dy, xw, gw, False)
    gw2 = torch.empty_like(gw)
    ops.wgrad(dy, xw, gw2, False)
    assert torch.equal(gw, gw2)
    assert rel_err(gw.cpu().numpy(), (dy.float().cpu().t() @ xw.float().cpu()).numpy()) < 1e-5


@pytest.mark.parametrize("case", ["s1_256", "s1_512", "s2", "up", "1x1", "s2_128", "s2_128_one_round", "up_128", "s1_64out",
                                  "s1_128_ragged", "1x1_k128", "conv_out_64_tiles"])
def test_conv_on_wide_kernel_at_size(case):
    """implicit-GEMM convolutions big enough for the persistent 256-wide kernel (im2col rows fetched by LDS-DMA with
    address predicates for the padding / stride-2 pad / nearest x2 upsampling) == F.conv2d on CPU."""
    from melspec_gpt_vqvae_amd import ops

    torch.manual_seed(11)
    if case == "s1_256":
        B, H, W, Cin, Cout, k, stride, up = 6, 40, 212, 256, 256, 3, 1, False
    elif case == "s1_512":
        B, H, W, Cin, Cout, k, stride, up = 48, 20, 53, 512, 512, 3, 1, False
    elif case == "s2":
        B, H, W, Cin, Cout, k, stride, up = 6, 80, 424, 256, 256, 3, 2, False
    elif case == "up":
        B, H, W, Cin, Cout, k, stride, up = 4, 40, 106, 256, 256, 3, 1, True
    elif case == "1x1":
        B, H, W, Cin, Cout, k, stride, up = 10, 40, 212, 256, 512, 1, 1, False
    # at most 128 output channels: 256 x 128 tiles of the ping-pong loop (csrc/gemm8p.hip, NHALF) - several rounds of the
    # block lists, one round (fewer tiles than CUs), nearest x2 upsampling, half the tile's columns, a ragged last tile
    elif case == "s2_128":
        B, H, W, Cin, Cout, k, stride, up = 6, 80, 848, 128, 128, 3, 2, False
    elif case == "s2_128_one_round":
        B, H, W, Cin, Cout, k, stride, up = 3, 80, 848, 128, 128, 3, 2, False
    elif case == "up_128":
        B, H, W, Cin, Cout, k, stride, up = 4, 40, 212, 256, 128, 3, 1, True
    elif case == "1x1_k128":      # two K tiles per output tile (the 128 -> 256 nin_shortcut at 20 x 212)
        B, H, W, Cin, Cout, k, stride, up = 10, 40, 212, 128, 256, 1, 1, False
    elif case == "conv_out_64_tiles":   # Encoder.conv_out at 64 tiles: 89 tiles of 192 x 256, ONE partial round of the persistent kernel
        B, H, W, Cin, Cout, k, stride, up = 64, 5, 53, 512, 256, 3, 1, False
    elif case == "s1_64out":
        B, H, W, Cin, Cout, k, stride, up = 4, 80, 424, 128, 64, 3, 1, False
    else:
        B, H, W, Cin, Cout, k, stride, up = 5, 77, 431, 64, 104, 3, 1, False
    x = (torch.randn(B, H, W, Cin) * 0.5).to(torch.bfloat16)
    w = (torch.randn(Cout, Cin, k, k) * 0.05).to(torch.bfloat16)
    bias = torch.randn(Cout) * 0.1
    xin = x.float().permute(0, 3, 1, 2)
    if up:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    if stride == 2:
        ref = F.conv2d(F.pad(xin, (0, 1, 0, 1)), w.float(), bias, stride=2)       # Downsample: pad (0,1,0,1), no padding
        pad = (0, 0)
    else:
        ref = F.conv2d(xin, w.float(), bias, padding=k // 2)
        pad = (k // 2, k // 2)
    ref = ref.permute(0, 2, 3, 1)
    res = torch.randn(ref.shape).to(torch.bfloat16)
    wp = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = ops.conv2d_nhwc(x.to(DEV), wp, bias.to(DEV), stride=stride, pad=pad, out_hw=tuple(ref.shape[1:3]), upsample=up,
                        residual=res.to(DEV))
    assert y.shape == ref.shape
    assert rel_err(y.float().cpu().numpy(), (ref + res.float()).numpy()) < 2 ** -8


@pytest.mark.parametrize("B", [130, 125])
def test_downsample_conv_over_two_gib_of_input_runs_as_whole_image_launches(B):
    """The 256 x 128-tile form addresses its operand with 32-bit offsets that end at 2 GiB; the training step's Downsample
    conv at 128 tiles of 80 x 848 x 128 reads 2.2 GB: melgpt_conv2d_nhwc cuts the batch into whole-image launches (B = 125:
    an UNEVEN cut, 63 + 62).  The result must equal the two parts run on their own (bit for bit) and F.conv2d on the first
    image, the two images either side of the cut and the last one."""
    from melspec_gpt_vqvae_amd import ops

    H, W, C = 80, 848, 128
    g = torch.Generator(device=DEV).manual_seed(19)
    x = (torch.randn(B, H, W, C, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    assert x.numel() * 2 > 2 ** 31
    w = (torch.randn(C, C, 3, 3, generator=torch.Generator().manual_seed(20)) * 0.05).to(torch.bfloat16)
    bias = torch.randn(C, generator=torch.Generator().manual_seed(21)) * 0.1
    wp = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = ops.conv2d_nhwc(x, wp, bias.to(DEV), stride=2, pad=(0, 0), out_hw=(40, 424))
    assert y.shape == (B, 40, 424, C)
    cut = (B + 1) // 2
    for lo, hi in ((0, cut), (cut, B)):
        assert torch.equal(y[lo:hi], ops.conv2d_nhwc(x[lo:hi].contiguous(), wp, bias.to(DEV), stride=2, pad=(0, 0), out_hw=(40, 424)))
    for b in (0, cut - 1, cut, B - 1):
        xin = F.pad(x[b:b + 1].float().cpu().permute(0, 3, 1, 2), (0, 1, 0, 1))
        ref = F.conv2d(xin, w.float(), bias, stride=2).permute(0, 2, 3, 1)
        assert rel_err(y[b:b + 1].float().cpu().numpy(), ref.numpy()) < 2 ** -8


@pytest.mark.parametrize("ak,bk,M,N,K", [(False, False, 33920, 1024, 1024), (False, True, 8192, 1024, 4096),
                                         (True, True, 4096, 1024, 8480)])
def test_wide_kernel_is_bit_reproducible(ak, bk, M, N, K):
    """The persistent kernel's LDS-DMA ring is ordered by hand (inline-asm pieces, counted waits, one barrier per K
    unit): a missing wait would read a half-landed tile - noise from run to run.  Twenty repeats must be bit-identical
    and agree with an f32 product of the same bf16 operands."""
    from melspec_gpt_vqvae_amd import ops

    torch.manual_seed(5)
    a = torch.randn((K, M) if ak else (M, K), device=DEV).bfloat16()
    b = torch.randn((K, N) if bk else (N, K), device=DEV).bfloat16()
    ref = ops.gemm(a, b, a_kmajor=ak, b_kmajor=bk)
    for _ in range(20):
        assert torch.equal(ops.gemm(a, b, a_kmajor=ak, b_kmajor=bk), ref)
    r32 = (a.float().T if ak else a.float()) @ (b.float() if bk else b.float().T)
    assert float((ref.float() - r32).abs().max() / r32.abs().max()) < 6e-3


def test_reserved_cus_leave_room_for_rccl_and_do_not_change_results():
    """melgpt_set_reserved_cus(n): the persistent GEMM / wide conv run on (CUs - n) workgroups (dp.DataParallel sets it
    when all-reduces overlap the backward pass); same bits as the full-chip launch."""
    from melspec_gpt_vqvae_amd import _ffi, ops

    torch.manual_seed(4)
    a = torch.randn(33920, 1024, device=DEV).bfloat16()
    b = torch.randn(1024, 1024, device=DEV).bfloat16()
    x = (torch.randn(8, 80, 848, 128, device=DEV) * 0.5).bfloat16()
    w = (torch.randn(128, 3, 3, 128, device=DEV) * 0.05).bfloat16()
    bias = torch.randn(128, device=DEV) * 0.1
    g, bt = torch.ones(128, device=DEV), torch.zeros(128, device=DEV)
    mean, rstd = ops.groupnorm_stats(x, 1e-6)
    ref = ops.gemm(a, b)
    cref = ops.conv3x3_gn(x, (mean, rstd), g, bt, w, bias)
    try:
        _ffi.call("melgpt_set_reserved_cus", 16)
        assert _ffi.lib().melgpt_get_reserved_cus() == 16
        assert torch.equal(ops.gemm(a, b), ref)
        assert torch.equal(ops.conv3x3_gn(x, (mean, rstd), g, bt, w, bias), cref)
        with pytest.raises(_ffi.MelgptError):
            _ffi.call("melgpt_set_reserved_cus", -1)
    finally:
        _ffi.call("melgpt_set_reserved_cus", 0)
    assert torch.equal(ops.gemm(a, b), ref)


def test_ring_loop_gives_the_same_bits_as_the_pingpong_loop_with_and_without_reserved_cus():
    """melgpt_set_gemm_pingpong(0): the persistent GEMM's ring K loop (csrc/gemm256.hip) behind the same tile lists, operand
    layouts and epilogues as the ping-pong loop (csrc/gemm8p.hip, the default).  Same tiles, same arithmetic: results must be
    bit-identical, launch after launch, for every operand layout / epilogue family / tile height - on the whole chip and
    with 16 CUs reserved (what a data-parallel run with a pinned RCCL channel count does during the backward pass)."""
    from melspec_gpt_vqvae_amd import _ffi, ops

    torch.manual_seed(8)
    a = torch.randn(33920, 1024, device=DEV).bfloat16()
    w = (torch.randn(4096, 1024, device=DEV) * 0.1).bfloat16()
    w2 = (torch.randn(1024, 4096, device=DEV) * 0.1).bfloat16()
    bias = torch.randn(4096, device=DEV) * 0.1
    r = torch.randn(33920, 1024, device=DEV).bfloat16()
    dy = (torch.randn(33920, 1024, device=DEV) * 0.1).bfloat16()

    def run():
        dact = torch.empty(33920, 4096, dtype=torch.bfloat16, device=DEV)
        act = ops.gemm(a, w, bias=bias, act=ops.ACT_GELU_DACT, pre_out=dact)                    # NT, EPI_DACT16
        y = ops.gemm(act, w2, bias=bias[:1024].contiguous(), residual=r, drop_p=0.5, seed=3, stream_id=1)  # NT, full mode
        g = ops.gemm(dy, w2, b_kmajor=True, act=ops.ACT_MUL, residual=dact)                     # NN, plain + R, 192 rows
        gw = torch.empty(4096, 1024, device=DEV)
        ops.wgrad(g, a, gw, False)                                                              # TN split-K, f32 out
        lg = ops.gemm(a, w[:128].contiguous(), out_dtype=torch.float32)                         # few tiles (< one round)
        sq = ops.gemm(a[:3584], w[:3584].contiguous())            # 196 tiles < 256 workgroups: one counter, linear order
        return act, dact, y, g, gw, lg, sq

    try:
        # 16 reserved CUs: 240 workgroups, 30 per XCD (the weight gradient's split-K factor follows the workgroup count, so
        # the reference is taken at the same reservation)
        for reserve, rounds in ((0, 3), (16, 2)):
            _ffi.call("melgpt_set_reserved_cus", reserve)
            _ffi.call("melgpt_set_gemm_pingpong", 1)
            ref = run()
            _ffi.call("melgpt_set_gemm_pingpong", 0)
            assert _ffi.lib().melgpt_get_gemm_pingpong() == 0
            for _ in range(rounds):
                for x, y in zip(run(), ref):
                    assert torch.equal(x, y)
    finally:
        _ffi.call("melgpt_set_gemm_pingpong", 1)
        _ffi.call("melgpt_set_reserved_cus", 0)


@pytest.mark.parametrize("N,K,M", [(1024, 1024, 33920), (4096, 1024, 33920), (1024, 4096, 33920), (3072, 1024, 16960),
                                    (256, 192, 2120)])
def test_weight_gradient_launch_also_yields_the_bias_gradient(N, K, M):
    """ops.wgrad(dy, x, dW, acc, bias_out=db): on the persistent kernel the column sums of dy (the bias gradient of
    y = x W^T + b) come out of the weight-gradient GEMM's own K loop (melgpt_wgrad_rowsum: one more MFMA per A fragment
    against a fragment of ones); dW is bit-identical to the launch without it, db equals the f64 column sums to f32
    accumulation accuracy, accumulate flags are honoured separately.  The small shape takes the melgpt_colsum path."""
    from melspec_gpt_vqvae_amd import ops

    torch.manual_seed(N + K)
    dy = (torch.randn(M, N, device=DEV) * 0.1 + 0.01).bfloat16()
    x = torch.randn(M, K, device=DEV).bfloat16()
    ref_w = torch.empty(N, K, device=DEV)
    ops.wgrad(dy, x, ref_w, False)
    w = torch.empty(N, K, device=DEV)
    b = torch.full((N,), 3.0, device=DEV)
    ops.wgrad(dy, x, w, False, bias_out=b, bias_accumulate=True)
    assert torch.equal(w, ref_w)
    want = dy.double().sum(0).cpu().numpy()
    got = (b - 3.0).double().cpu().numpy()
    assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max())
    ops.wgrad(dy, x, w, True, bias_out=b, bias_accumulate=False)        # weights accumulate, bias overwritten
    assert rel_err(w.cpu().numpy(), (2 * ref_w).cpu().numpy()) < 1e-6
    assert np.abs(b.double().cpu().numpy() - want).max() < 2e-5 * max(1.0, np.abs(want).max())
