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
