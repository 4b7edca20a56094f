"""MelGAN generator (melspec_gpt_vqvae_amd/vocoder/modules.py) against outputs recorded from the real
reference Generator (vocoder/modules.py:38-79) with seeded weight-normed weights (tests/golden/melgan_small.npz).
f32 lane gate 1e-4 relative to max; bf16 lane reported with a loose bound."""
import numpy as np
import pytest
import torch

import synth
from util import report, golden, rel_err, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _gen(g):
    from melspec_gpt_vqvae_amd.vocoder import Generator

    m = Generator(80, int(g["ngf"]), 3)
    sd = synth.melgan_state_dict(int(g["seed"]), input_size=80, ngf=int(g["ngf"]), n_residual_layers=3)
    assert [k for k in m.state_dict().keys()] == [str(k) for k in g["sd_keys"]] == list(sd.keys()), "checkpoint ABI"
    res = m.load_state_dict({k: t(v) for k, v in sd.items()})
    assert not res.missing_keys and not res.unexpected_keys
    return m.to(DEV).eval()


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_generator_matches_reference_golden(dt):
    g = golden("melgan_small")
    m = _gen(g)
    if dt == "bf16":
        for mod in m.modules():
            object.__setattr__(mod, "compute_dtype", torch.bfloat16)
    y = m(t(g["x"], DEV))
    assert y.shape == (2, 1, 12 * 256) and y.dtype == torch.float32 and int(m.hop_length) == 256
    err = rel_err(y.cpu().numpy(), g["y"])
    report("melgan_vs_reference", lane=dt, rel_to_max_err=err)
    assert err < (1e-4 if dt == "f32" else 6e-2)
    assert float(y.abs().max()) <= 1.0                       # tanh range


def test_resnet_block_and_reflect_padding_pieces():
    import torch.nn.functional as F

    from melspec_gpt_vqvae_amd import ops

    g = golden("melgan_small")
    m = _gen(g)
    yb = m.model[4](t(g["xb"], DEV))                         # (B, C, L) in and out, like the reference
    assert rel_err(yb.cpu().numpy(), g["yb"]) < 1e-4
    # padding copy: reflection / zero rows, LeakyReLU(0.2) on the way
    x = t(synth.normal(5, (2, 9, 16), 1.0))
    for pad, reflect, slope in ((3, True, 1.0), (9 - 1, True, 0.2), (1, False, 0.2), (0, True, 0.2)):
        y = ops.pad1d_act(x.to(DEV), pad, reflect=reflect, slope=slope)
        xa = F.leaky_relu(x, slope) if slope != 1.0 else x
        xc = xa.permute(0, 2, 1)
        ref = (F.pad(xc, (pad, pad), mode="reflect") if reflect and pad else F.pad(xc, (pad, pad))).permute(0, 2, 1)
        assert torch.equal(y.cpu(), ref)
    # last layer: 7 taps to one channel + tanh
    C, L = 8, 33
    xp = t(synth.normal(6, (2, L + 6, C), 1.0))
    w = t(synth.normal(7, (1, C, 7), 0.2))
    b = t(synth.normal(8, (1,), 0.1))
    y = ops.conv1d_out1(xp.to(DEV), w[0].t().contiguous().reshape(-1).to(DEV), b.to(DEV), L, 7)
    ref = torch.tanh(F.conv1d(xp.permute(0, 2, 1), w, b))[:, 0]
    assert rel_err(y.cpu().numpy(), ref.numpy()) < 1e-5


@pytest.mark.parametrize("cin,cout,k,dil,reflect,slope", [(80, 64, 7, 1, True, 0.0), (32, 32, 3, 9, True, 0.2),
                                                           (64, 64, 3, 3, True, 0.2), (128, 128, 3, 1, True, 0.2),
                                                           (32, 48, 1, 1, False, 0.2), (64, 32, 2, 1, False, 0.2)])
def test_conv1d_as_one_implicit_gemm_matches_torch(cin, cout, k, dil, reflect, slope):
    """melgpt_conv1d_nlc, f32 lane, against F.conv1d on the explicitly activated + padded input: taps as K steps,
    reflection as an address predicate, LeakyReLU on the operand, Cin not a multiple of the K step (a lane's chunk carries
    its own tap), accumulate and a row-strided output (vocoder/modules.py:23-79)."""
    import torch.nn.functional as F

    from melspec_gpt_vqvae_amd import ops

    B, L = 2, 40
    x = t(synth.normal(11, (B, L, cin), 1.0))
    w = t(synth.normal(12, (cout, cin, k), 0.1))
    b = t(synth.normal(13, (cout,), 0.1))
    pad = dil * (k - 1) // 2 if k % 2 else 1
    xa = F.leaky_relu(x, slope) if slope else x
    xc = xa.permute(0, 2, 1)
    right = dil * (k - 1) - pad
    xp = F.pad(xc, (pad, right), mode="reflect") if reflect else F.pad(xc, (pad, right))
    ref = F.conv1d(xp, w, b, dilation=dil).permute(0, 2, 1)                       # (B, L, cout)
    wcat = w.permute(0, 2, 1).reshape(cout, -1).contiguous().to(DEV)
    y = ops.conv1d_nlc(x.to(DEV), wcat, b.to(DEV), k, dilation=dil, pad_l=pad, reflect=reflect, in_slope=slope)
    assert rel_err(y.cpu().numpy(), ref.numpy()) < 1e-5
    yl = ops.conv1d_nlc(x.to(DEV), wcat, b.to(DEV), k, dilation=dil, pad_l=pad, reflect=reflect, in_slope=slope, out_slope=0.2)
    assert rel_err(yl.cpu().numpy(), F.leaky_relu(ref, 0.2).numpy()) < 1e-5       # LeakyReLU on the output (conv3 -> conv1)
    # accumulate into every second row of a wider buffer (a transposed convolution's output phase)
    big = t(synth.normal(14, (B, 2 * L, cout), 1.0)).to(DEV)
    keep = big.clone()
    ops.conv1d_nlc(x.to(DEV), wcat, b.to(DEV), k, dilation=dil, pad_l=pad, reflect=reflect, in_slope=slope, out=big[:, 1::2, :],
                   accumulate=True)
    assert torch.equal(big[:, 0::2, :], keep[:, 0::2, :])
    assert rel_err((big[:, 1::2, :] - keep[:, 1::2, :]).cpu().numpy(), ref.numpy()) < 1e-5


@pytest.mark.parametrize("r,cin,cout", [(8, 64, 32), (2, 32, 16)])
def test_conv_transpose_phases_match_torch(r, cin, cout):
    import torch.nn as nn
    import torch.nn.functional as F

    from melspec_gpt_vqvae_amd.vocoder import modules as vm

    torch.manual_seed(3)
    m = vm.WNConvTranspose1d(cin, cout, kernel_size=2 * r, stride=r, padding=r // 2 + r % 2, output_padding=r % 2).to(DEV)
    x = torch.randn(2, 24, cin, device=DEV)
    y = vm._conv_transpose1d(x, m)                                               # LeakyReLU -> ConvTranspose1d, channels-last
    ref = m(F.leaky_relu(x, 0.2).permute(0, 2, 1)).permute(0, 2, 1)
    assert y.shape == ref.shape and rel_err(y.detach().cpu().numpy(), ref.detach().cpu().numpy()) < 1e-5


@pytest.mark.parametrize("C,dil,L", [(32, 1, 64), (32, 9, 48), (64, 3, 64), (64, 9, 32), (128, 1, 64), (128, 9, 128), (128, 3, 192)])
def test_narrow_resnet_block_in_one_pass_equals_the_three_convolutions(C, dil, L):
    """resblock_narrow_kernel (weights as register fragments, t1 never stored) against the same block run as three
    implicit GEMMs on the 16-bit lane: same operands, t1 rounded to the 16-bit format in both; and against torch in f32."""
    import torch.nn.functional as F

    from melspec_gpt_vqvae_amd import _ffi, ops
    from melspec_gpt_vqvae_amd.vocoder import modules as vm

    torch.manual_seed(C + dil)
    blk = vm.ResnetBlock(C, dilation=dil).to(DEV).eval()
    x = torch.randn(3, L, C, device=DEV)                                         # three clips: reflections at every clip end
    xh = x.to(_ffi.HALF_DTYPE)
    y1 = blk._run(xh)                                                           # one pass (C in {32, 64}, L % 16 == 0)
    d = blk.block[1].padding[0]
    t1 = vm._conv1d(xh, blk.block[2], pad=d, leaky=True)
    y3 = vm._conv1d(xh, blk.shortcut, pad=0)
    y3 = vm._conv1d(t1, blk.block[4], pad=0, leaky=True, out=y3, accumulate=True)
    ref = (blk.shortcut(xh.float().permute(0, 2, 1)) + blk.block(xh.float().permute(0, 2, 1))).permute(0, 2, 1)
    e13 = rel_err(y1.float().cpu().numpy(), y3.float().cpu().numpy())
    e1r = rel_err(y1.float().cpu().numpy(), ref.detach().cpu().numpy())
    report("melgan_resblock_one_pass", C=C, dilation=dil, vs_three_convs=e13, vs_torch_f32=e1r)
    assert e13 < 1.5e-2 and e1r < 1.5e-2


def test_output_layer_in_one_pass_matches_torch():
    import torch.nn.functional as F

    from melspec_gpt_vqvae_amd import _ffi, ops

    for C in (32, 64):
        B, L, K = 2, 300, 7
        h = t(synth.normal(21, (B, L, C), 1.0)).to(DEV).to(_ffi.HALF_DTYPE)
        w = t(synth.normal(22, (1, C, K), 0.1))
        b = t(synth.normal(23, (1,), 0.1))
        y = ops.conv1d_out1_fused(h, w[0].t().contiguous().reshape(-1).to(DEV), b.to(DEV), K, 0.2)
        xa = F.leaky_relu(h.float().cpu(), 0.2).to(_ffi.HALF_DTYPE).float()      # the kernel rounds the activation to 16 bits
        ref = torch.tanh(F.conv1d(F.pad(xa.permute(0, 2, 1), (3, 3), mode="reflect"), w, b))[:, 0]
        assert rel_err(y.cpu().numpy(), ref.numpy()) < 1e-5
