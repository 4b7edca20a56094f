"""GPU parity of the VQ-VAE backward exports (melgpt_conv3x3_bwd_data / _bwd_weight, melgpt_groupnorm_swish_bwd;
csrc/vqvae_bwd.hip, through the C ABI) against (1) gradients recorded from the REAL reference's ResnetBlock
(tests/golden/resblock_grad.npz, made by tests/golden/make_golden.py with torch autograd on the imported module) and (2) plain fp32
PyTorch autograd on CPU at other shapes.  f32 lane gate 1e-4 (relative to the tensor's maximum); the 16-bit lane is reported against
the same references at its own resolution.  The exports are not wired into LitVQVAE's autograd: no scored configuration trains the
VQ-VAE, and .backward() through the module is still refused (tests/test_vqvae_gpu.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import synth
from util import golden, rel_err, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DT = {"f32": torch.float32, "bf16": torch.bfloat16}
TOL = {"f32": 1e-4, "bf16": 3e-2}


def nhwc(a):      # (B,C,H,W) numpy -> (B,H,W,C) contiguous tensor
    return torch.from_numpy(np.ascontiguousarray(np.transpose(a, (0, 2, 3, 1))))


def nchw(x):      # (B,H,W,C) tensor -> (B,C,H,W) numpy f32
    return x.float().cpu().permute(0, 3, 1, 2).numpy()


def oihw(dw):     # packed (Cout,3,3,Cin) gradient -> the reference's (Cout,Cin,3,3)
    return dw.float().cpu().permute(0, 3, 1, 2).numpy()


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_resnet_block_backward_against_the_real_reference(dt):
    """forward of the block on the library's own kernels, backward composed of the three exports: y, dx and every parameter
    gradient against the fixture recorded from the reference's ResnetBlock (big_model_attn_gan.py:75-135)."""
    from melspec_gpt_vqvae_amd import ops

    g = golden("resblock_grad")
    C = int(g["channels"])
    sd = synth.resblock_state_dict(int(g["seed"]), C)
    dtp = DT[dt]
    x = nhwc(g["x"]).to(dtp).to(DEV)
    gy = nhwc(g["g"]).to(dtp).to(DEV)
    dev = lambda k: t(sd[k]).to(DEV)
    wp = {k: t(sd[k + ".weight"]).permute(0, 2, 3, 1).contiguous().to(dtp).to(DEV) for k in ("conv1", "conv2")}
    # ---- forward: norm1 -> swish -> conv1 -> norm2 -> swish -> conv2, + x
    s1 = ops.groupnorm_stats(x, 1e-6)
    a1 = ops.groupnorm(x, dev("norm1.weight"), dev("norm1.bias"), 1e-6, swish=True)
    h1 = ops.conv2d_nhwc(a1, wp["conv1"], dev("conv1.bias"))
    s2 = ops.groupnorm_stats(h1, 1e-6)
    a2 = ops.groupnorm(h1, dev("norm2.weight"), dev("norm2.bias"), 1e-6, swish=True)
    y = ops.conv2d_nhwc(a2, wp["conv2"], dev("conv2.bias"), residual=x)
    assert rel_err(nchw(y), g["y"]) < TOL[dt]
    # ---- backward
    da2 = ops.conv3x3_bwd_data(gy, wp["conv2"], C)
    dw2, db2 = ops.conv3x3_bwd_weight(a2, gy)
    dh1, dg2, dbt2 = ops.groupnorm_swish_bwd(h1, s2, dev("norm2.weight"), dev("norm2.bias"), da2)
    da1 = ops.conv3x3_bwd_data(dh1, wp["conv1"], C)
    dw1, db1 = ops.conv3x3_bwd_weight(a1, dh1)
    dxb, dg1, dbt1 = ops.groupnorm_swish_bwd(x, s1, dev("norm1.weight"), dev("norm1.bias"), da1)
    dx = nchw(dxb) + g["g"]                                  # the shortcut's share of the gradient (out = x + h)
    tol = TOL[dt]
    errs = {"dx": rel_err(dx, g["dx"]),
            "d_conv2_weight": rel_err(oihw(dw2), g["d_conv2_weight"]), "d_conv2_bias": rel_err(db2.cpu().numpy(), g["d_conv2_bias"]),
            "d_norm2_weight": rel_err(dg2.cpu().numpy(), g["d_norm2_weight"]), "d_norm2_bias": rel_err(dbt2.cpu().numpy(), g["d_norm2_bias"]),
            "d_conv1_weight": rel_err(oihw(dw1), g["d_conv1_weight"]), "d_conv1_bias": rel_err(db1.cpu().numpy(), g["d_conv1_bias"]),
            "d_norm1_weight": rel_err(dg1.cpu().numpy(), g["d_norm1_weight"]), "d_norm1_bias": rel_err(dbt1.cpu().numpy(), g["d_norm1_bias"])}
    assert all(v < tol for v in errs.values()), errs


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 7, 13, 64, 128), (1, 20, 53, 128, 128), (2, 5, 53, 128, 64), (5, 9, 4, 64, 64)])
def test_conv3x3_backward_against_torch_autograd(dt, B, H, W, Cin, Cout):
    from melspec_gpt_vqvae_amd import ops

    dtp = DT[dt]
    x = t(synth.normal(71, (B, Cin, H, W), 1.0, 0.2)).to(dtp)
    w = t(synth.normal(72, (Cout, Cin, 3, 3), 0.05)).to(dtp)
    gy = t(synth.normal(73, (B, Cout, H, W), 0.8)).to(dtp)
    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    b = torch.zeros(Cout, requires_grad=True)
    F.conv2d(xr, wr, b, padding=1).backward(gy.float())
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    gd = gy.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    dx = ops.conv3x3_bwd_data(gd, wp, Cin)
    dw, db = ops.conv3x3_bwd_weight(xd, gd)
    tol = TOL[dt]
    assert rel_err(nchw(dx), xr.grad.numpy()) < tol
    assert rel_err(oihw(dw), wr.grad.numpy()) < tol
    assert rel_err(db.cpu().numpy(), b.grad.numpy()) < tol
    # deterministic: the same bits twice (fixed-order partial sums)
    dw2, db2 = ops.conv3x3_bwd_weight(xd, gd)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("swish", [True, False])
@pytest.mark.parametrize("B,HW,C", [(2, (5, 53), 512), (3, (20, 31), 128), (1, (40, 33), 256), (4, (3, 3), 64)])
def test_groupnorm_swish_backward_against_torch_autograd(dt, swish, B, HW, C):
    from melspec_gpt_vqvae_amd import ops

    dtp = DT[dt]
    x = t(synth.normal(81, (B, C, HW[0], HW[1]), 1.4, 0.6)).to(dtp)
    gy = t(synth.normal(82, (B, C, HW[0], HW[1]), 0.9)).to(dtp)
    gm, bt = t(synth.normal(83, (C,), 0.1, 1.0)), t(synth.normal(84, (C,), 0.1))
    xr, gr, br = x.float().requires_grad_(True), gm.clone().requires_grad_(True), bt.clone().requires_grad_(True)
    h = F.group_norm(xr, 32, gr, br, eps=1e-6)
    (h * torch.sigmoid(h) if swish else h).backward(gy.float())
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    gd = gy.permute(0, 2, 3, 1).contiguous().to(DEV)
    stats = ops.groupnorm_stats(xd, 1e-6)
    dx, dg, db = ops.groupnorm_swish_bwd(xd, stats, gm.to(DEV), bt.to(DEV), gd, swish=swish)
    tol = TOL[dt]
    assert rel_err(nchw(dx), xr.grad.numpy()) < tol
    assert rel_err(dg.cpu().numpy(), gr.grad.numpy()) < tol and rel_err(db.cpu().numpy(), br.grad.numpy()) < tol


def test_conv3x3_backward_at_the_reference_layer_size():
    """the layer that decides the encoder (128 -> 128 at 80 x 848, big_model_attn_gan.py:203-251), two tiles, 16-bit lane: the
    weight gradient reduces over 2 x 82 x 850 zero-bordered rows in 68 split-K batches; against torch autograd in f32 on the
    same 16-bit-rounded operands."""
    from melspec_gpt_vqvae_amd import ops

    B, H, W, C = 2, 80, 848, 128
    x = t(synth.normal(91, (B, C, H, W), 1.0, 0.2)).to(torch.bfloat16)
    w = t(synth.normal(92, (C, C, 3, 3), 0.03)).to(torch.bfloat16)
    gy = t(synth.normal(93, (B, C, H, W), 0.05)).to(torch.bfloat16)
    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    b = torch.zeros(C, requires_grad=True)
    F.conv2d(xr, wr, b, padding=1).backward(gy.float())
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    gd = gy.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    dx = ops.conv3x3_bwd_data(gd, wp, C)
    dw, db = ops.conv3x3_bwd_weight(xd, gd)
    assert rel_err(nchw(dx), xr.grad.numpy()) < 8e-3          # (dx is stored in 16 bits)
    assert rel_err(oihw(dw), wr.grad.numpy()) < 1e-4          # (f32 accumulation of exact 16-bit products, f32 output)
    assert rel_err(db.cpu().numpy(), b.grad.numpy()) < 1e-4


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 10, 22, 64, 64), (3, 20, 106, 128, 128), (1, 8, 6, 64, 128)])
def test_downsample_conv_backward_against_torch_autograd(dt, B, H, W, Cin, Cout):
    """Downsample: F.pad (0,1,0,1) + conv3x3 stride 2 (big_model_attn_gan.py:151-159)."""
    from melspec_gpt_vqvae_amd import ops

    dtp = DT[dt]
    OH, OW = (H - 2) // 2 + 1, (W - 2) // 2 + 1
    x = t(synth.normal(101, (B, Cin, H, W), 1.0, 0.2)).to(dtp)
    w = t(synth.normal(102, (Cout, Cin, 3, 3), 0.05)).to(dtp)
    gy = t(synth.normal(103, (B, Cout, OH, OW), 0.8)).to(dtp)
    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    b = torch.zeros(Cout, requires_grad=True)
    y = F.conv2d(F.pad(xr, (0, 1, 0, 1)), wr, b, stride=2)
    assert y.shape[2:] == (OH, OW)
    y.backward(gy.float())
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    gd = gy.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    dx, dw, db = ops.conv3x3_s2_bwd(xd, gd, wp)
    tol = TOL[dt]
    assert rel_err(nchw(dx), xr.grad.numpy()) < tol
    assert rel_err(oihw(dw), wr.grad.numpy()) < tol
    assert rel_err(db.cpu().numpy(), b.grad.numpy()) < tol


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_upsample_conv_backward_against_torch_autograd(dt):
    """Upsample: nearest x2 + conv3x3 (:171-186) = the stride-1 exports on the x2 geometry + upsample2 / its adjoint sumpool2."""
    from melspec_gpt_vqvae_amd import ops

    dtp = DT[dt]
    B, H, W, C = 2, 5, 13, 64
    x = t(synth.normal(111, (B, C, H, W), 1.0, 0.2)).to(dtp)
    w = t(synth.normal(112, (C, C, 3, 3), 0.05)).to(dtp)
    gy = t(synth.normal(113, (B, C, 2 * H, 2 * W), 0.8)).to(dtp)
    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    F.conv2d(F.interpolate(xr, scale_factor=2.0, mode="nearest"), wr, None, padding=1).backward(gy.float())
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    gd = gy.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    xu = ops.upsample2(xd)
    assert torch.equal(xu.float().cpu().permute(0, 3, 1, 2), F.interpolate(x.float(), scale_factor=2.0, mode="nearest"))
    dxu = ops.conv3x3_bwd_data(gd, wp, C)
    dx = ops.sumpool2(dxu)
    dw, _ = ops.conv3x3_bwd_weight(xu, gd, want_bias=False)
    tol = TOL[dt]
    assert rel_err(nchw(dx), xr.grad.numpy()) < tol
    assert rel_err(oihw(dw), wr.grad.numpy()) < tol


def _narrow_litvqvae():
    """the package's LitVQVAE with the module globals patched to the fixture's narrow sizes (ch = 32, z_channels = 64)"""
    import melspec_gpt_vqvae_amd.vqvae.big_model_attn_gan as vq

    saved = {k: getattr(vq, k) for k in ("ch", "z_channels")}
    try:
        vq.ch, vq.z_channels = 32, 64
        m = vq.LitVQVAE(num_embeddings=128, embedding_dim=256)
    finally:
        for k, v in saved.items():
            setattr(vq, k, v)
    sd = synth.vqvae_state_dict(70, num_embeddings=128, embedding_dim=256, ch=32, z_channels=64)
    res = m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    assert all(k.startswith("discriminator.") for k in res.missing_keys) and not res.unexpected_keys
    return m.to(DEV).train()


def test_litvqvae_forward_is_differentiable_like_the_reference():
    """LitVQVAE.forward (big_model_attn_gan.py:622-634) end to end on the f32 lane - encoder, quant_conv, VectorQuantizer with
    its straight-through estimator, post_quant_conv, decoder - against the gradients of the REAL module recorded by
    tests/golden/make_golden.py (narrow model, one 80 x 848 tile; L = vq_loss + sum(x_recon * g)): the loss, the codes, the
    reconstruction, and for EVERY one of the 343 parameters the gradient's norm and 16 sampled entries, 1e-4 of the tensor's scale."""
    g = golden("vqvae_grad")
    m = _narrow_litvqvae()
    x = t(2 * synth.mel_tiles(71, 1)[:, None, :, 6:854] - 1, DEV)
    gy = t(synth.normal(72, (1, 1, 80, 848), 0.01), DEV)
    loss, x_recon, info = m(x)
    assert x_recon.shape == x.shape and x_recon.requires_grad and loss.requires_grad
    assert np.array_equal(info[2].cpu().numpy().ravel().astype(np.int16), g["indices"])
    L = loss + (x_recon.float() * gy).sum()
    assert abs(float(loss.detach()) - float(g["vq_loss"])) < 1e-4 * abs(float(g["vq_loss"]))
    assert abs(float(L.detach()) - float(g["L"])) < 1e-4 * max(1.0, abs(float(g["L"])))
    assert rel_err(x_recon.detach()[0, 0, 30:34, 400:408].cpu().numpy(), g["rec_patch"]) < 1e-4
    L.backward()
    params = dict(m.named_parameters())
    names = [str(n) for n in g["names"]]
    S = float(np.median([float(g["n__" + n.replace(".", "__")]) for n in names]))      # typical gradient norm (1.4)
    bad, zeros = {}, 0
    for name in names:
        key = name.replace(".", "__")
        p = params[name]
        assert p.grad is not None, name
        flat = p.grad.detach().double().cpu().numpy().ravel()
        nrm, samp, pos = float(g["n__" + key]), g["s__" + key].astype(np.float64), g["p__" + key]
        ours = float(np.sqrt((flat * flat).sum()))
        if nrm < 1e-4 * S:
            # analytically ZERO gradients - the key bias of an AttnBlock (softmax is invariant to it) and, at one channel per
            # GroupNorm group (ch = 32), a conv bias in front of a norm: the reference holds f32 noise (1e-8 .. 2e-5), so must we
            zeros += 1
            if ours >= 1e-4 * S:
                bad[name] = ("zero-gradient parameter", ours)
            continue
        scale = nrm / np.sqrt(flat.size)                        # rms entry of the reference gradient
        e_n = abs(ours - nrm) / nrm
        e_s = float(np.abs(flat[pos] - samp).max() / max(np.abs(samp).max(), scale))
        if e_n > 1e-4 or e_s > 1e-3:
            bad[name] = (e_n, e_s)
    assert not bad, (len(bad), list(bad.items())[:8])
    assert zeros == 21 and len(names) == 343
    assert all(p.grad is None for n, p in params.items() if n.startswith("discriminator."))


def test_inference_path_is_untouched_by_autograd_and_frozen_modules_record_nothing():
    """torch.no_grad() (every production caller) and a frozen module fed a plain input keep the fused inference kernels;
    with autograd recording the differentiable (unfused) path gives the same result to f32 rounding."""
    m = _narrow_litvqvae()
    x = t(2 * synth.mel_tiles(73, 1)[:, None, :, 6:854] - 1, DEV)
    with torch.no_grad():
        z0 = m.encode(x)
    z = m.encode(x)
    assert z.requires_grad and rel_err(z.detach().cpu().numpy(), z0.cpu().numpy()) < 1e-4
    for p in m.parameters():
        p.requires_grad_(False)
    z1 = m.encode(x)
    assert not z1.requires_grad and torch.equal(z1, z0)


def test_backward_exports_refuse_what_they_do_not_serve():
    from melspec_gpt_vqvae_amd import _ffi, ops

    dy = torch.zeros(1, 4, 4, 48, dtype=torch.bfloat16, device=DEV)       # Cout = 48: not a whole MFMA contraction step
    w = torch.zeros(48, 3, 3, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(_ffi.MelgptError):
        ops.conv3x3_bwd_data(dy, w, 64)
    x = torch.zeros(1, 4, 4, 40, dtype=torch.float32, device=DEV)        # C = 40: not 32 groups
    st = (torch.zeros(32, device=DEV), torch.ones(32, device=DEV))
    with pytest.raises(_ffi.MelgptError):
        ops.groupnorm_swish_bwd(x, st, torch.ones(40, device=DEV), torch.zeros(40, device=DEV), x)
