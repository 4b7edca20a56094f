"""GPU parity of the VQ-VAE encoder / decoder modules (melspec_gpt_vqvae_amd/vqvae/big_model_attn_gan.py on the
HIP kernels) against golden vectors recorded from the real reference (tests/golden/vqvae_*.npz) and plain
fp32 PyTorch on CPU.  f32 lane gate 1e-4; codes bit-exact under the SURVEY §8a tie policy."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import synth
from util import report, check_indices_with_tie_policy, golden, rel_err, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DT = {"f32": torch.float32, "bf16": torch.bfloat16}


def _load(module, sd_np, allow_missing_prefix=None):
    res = module.load_state_dict({k: t(v) for k, v in sd_np.items()}, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys[:4]
    if allow_missing_prefix is None:
        assert not res.missing_keys, res.missing_keys[:4]
    else:
        assert all(k.startswith(allow_missing_prefix) for k in res.missing_keys), res.missing_keys[:4]
    return module


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("C,HW", [(128, (80, 848)), (512, (5, 53)), (256, (20, 212))])
def test_groupnorm_swish(dt, C, HW):
    from melspec_gpt_vqvae_amd import ops

    B = 2
    x = t(synth.normal(1, (B, HW[0], HW[1], C), 1.5, 0.7)).to(DT[dt])
    gm, bt = t(synth.normal(2, (C,), 0.1, 1.0)), t(synth.normal(3, (C,), 0.1))
    for swish in (True, False):
        y = ops.groupnorm(x.to(DEV), gm.to(DEV), bt.to(DEV), 1e-6, swish=swish)
        ref = F.group_norm(x.float().permute(0, 3, 1, 2), 32, gm, bt, eps=1e-6)
        if swish:
            ref = ref * torch.sigmoid(ref)
        ref = ref.permute(0, 2, 3, 1)
        assert rel_err(y.float().cpu().numpy(), ref.numpy()) < (2e-5 if dt == "f32" else 8e-3)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("C,HW", [(256, (10, 106)), (512, (5, 53)), (256, (5, 53)), (512, (10, 106)), (128, (5, 53)),
                                  (256, (3, 7)), (512, (9, 32)), (256, (17, 17)), (512, (24, 24)), (256, (1, 577)),
                                  (256, (32, 36)), (512, (1, 1))])
def test_groupnorm_in_one_launch_for_small_images(dt, C, HW):
    """melgpt_groupnorm_fused (statistics + normalisation out of registers, the tensor read once) against torch's
    group_norm, against the three-kernel path, and its optional mean / rstd outputs; every workgroup size's boundary
    (288 / 289, 576 / 577, 1152 pixels); a shape it must decline (bf16 with 4-channel groups: 8 bytes per group)."""
    from melspec_gpt_vqvae_amd import _ffi, ops
    from melspec_gpt_vqvae_amd.ops import ptr, dtype_code, stream

    B = 3
    x = t(synth.normal(21, (B, HW[0], HW[1], C), 1.5, 0.7)).to(DT[dt]).to(DEV)
    gm, bt = t(synth.normal(22, (C,), 0.1, 1.0)).to(DEV), t(synth.normal(23, (C,), 0.1)).to(DEV)
    L = _ffi.lib()
    for swish in (True, False):
        y = torch.empty_like(x)
        mean, rstd = torch.empty(B * 32, device=DEV), torch.empty(B * 32, device=DEV)
        code = L.melgpt_groupnorm_fused(ptr(x), ptr(gm), ptr(bt), ptr(y), B, HW[0] * HW[1], C, 1e-6, int(swish), ptr(mean),
                                        ptr(rstd), dtype_code(x.dtype), stream())
        if dt == "bf16" and C == 128:
            assert code == _ffi.ERR_UNSUPPORTED
            continue
        assert code == 0
        ref = F.group_norm(x.float().cpu().permute(0, 3, 1, 2), 32, gm.cpu(), bt.cpu(), eps=1e-6)
        if swish:
            ref = ref * torch.sigmoid(ref)
        ref = ref.permute(0, 2, 3, 1)
        tol = 2e-5 if dt == "f32" else 8e-3
        assert rel_err(y.float().cpu().numpy(), ref.numpy()) < tol
        m3, r3 = ops.groupnorm_stats(x, 1e-6)
        assert rel_err(mean.cpu().numpy(), m3.cpu().numpy()) < 1e-5 and rel_err(rstd.cpu().numpy(), r3.cpu().numpy()) < 1e-5
        y3 = torch.empty_like(x)
        _ffi.call("melgpt_groupnorm_apply", ptr(x), ptr(m3), ptr(r3), ptr(gm), ptr(bt), ptr(y3), B, HW[0] * HW[1], C, int(swish),
                  dtype_code(x.dtype), stream())
        assert rel_err(y.float().cpu().numpy(), y3.float().cpu().numpy()) < tol
        assert torch.equal(ops.groupnorm(x, gm, bt, 1e-6, swish=swish), y)      # ... and it is what ops.groupnorm runs
    # one pixel more than the largest workgroup holds: declined, nothing launched
    xb = torch.zeros(1, 1, 1153, C, dtype=DT[dt], device=DEV)
    assert L.melgpt_groupnorm_fused(ptr(xb), ptr(gm), ptr(bt), ptr(torch.empty_like(xb)), 1, 1153, C, 1e-6, 1, None, None,
                                    dtype_code(xb.dtype), stream()) == _ffi.ERR_UNSUPPORTED


@pytest.mark.parametrize("dt,C", [("bf16", 384), ("bf16", 640), ("bf16", 768), ("f32", 160), ("f32", 192)])
def test_groupnorm_fused_declines_groups_that_straddle_16_byte_pieces(dt, C):
    """Group sizes that are not 1, 2 or 4 whole 16-byte pieces (bf16 C = 384: 24-byte groups; 640 / 768: 40 / 48; f32
    C = 160 / 192: 20 / 24) pass the row-width test (C * es multiple of 128) but a piece would straddle two groups: the one-launch
    kernel must decline them with nothing launched, and ops.groupnorm must still be right on them (three-kernel path)."""
    from melspec_gpt_vqvae_amd import _ffi, ops
    from melspec_gpt_vqvae_amd.ops import ptr, dtype_code, stream

    B, HW = 2, (5, 53)
    x = t(synth.normal(31, (B, HW[0], HW[1], C), 1.5, 0.7)).to(DT[dt]).to(DEV)
    gm, bt = t(synth.normal(32, (C,), 0.1, 1.0)).to(DEV), t(synth.normal(33, (C,), 0.1)).to(DEV)
    y = torch.full_like(x, 7.0)
    code = _ffi.lib().melgpt_groupnorm_fused(ptr(x), ptr(gm), ptr(bt), ptr(y), B, HW[0] * HW[1], C, 1e-6, 1, None, None,
                                             dtype_code(x.dtype), stream())
    assert code == _ffi.ERR_UNSUPPORTED
    torch.cuda.synchronize()
    assert bool((y == 7.0).all()), "declined means nothing was launched"
    ref = F.group_norm(x.float().cpu().permute(0, 3, 1, 2), 32, gm.cpu(), bt.cpu(), eps=1e-6)
    ref = (ref * torch.sigmoid(ref)).permute(0, 2, 3, 1)
    try:
        got = ops.groupnorm(x, gm, bt, 1e-6, swish=True)
    except _ffi.MelgptError as e:       # the three-kernel path's own guard (256 % (C / vec) != 0) may refuse the width: loudly
        assert f"({_ffi.ERR_UNSUPPORTED})" in str(e)
        return
    assert rel_err(got.float().cpu().numpy(), ref.numpy()) < (2e-5 if dt == "f32" else 8e-3)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv_in_out_single_channel_and_permute(dt):
    from melspec_gpt_vqvae_amd import ops

    B, H, W, C = 2, 80, 848, 128
    x = t(2 * synth.mel_tiles(5, B)[:, :, 6:854] - 1)
    w = t(synth.uniform(6, (C, 1, 3, 3), -0.3, 0.3))
    b = t(synth.uniform(7, (C,), -0.3, 0.3))
    y, none = ops.conv_in_c1(x.to(DEV), w.to(DEV), b.to(DEV), DT[dt])
    assert none is None
    ref = F.conv2d(x[:, None], w, b, padding=1).permute(0, 2, 3, 1)
    assert rel_err(y.float().cpu().numpy(), ref.numpy()) < (1e-6 if dt == "f32" else 8e-3)
    # the same launch taking the GroupNorm(32) statistics of its output (what the first ResnetBlock's norm1 needs):
    # identical tensor, statistics = those of the stored values (the two-pass kernel's, to summation order)
    y2, (mean, rstd) = ops.conv_in_c1(x.to(DEV), w.to(DEV), b.to(DEV), DT[dt], stats_eps=1e-6)
    assert torch.equal(y2, y)
    m_ref, r_ref = ops.groupnorm_stats(y, 1e-6)
    assert rel_err(mean.cpu().numpy(), m_ref.cpu().numpy()) < 1e-5 and rel_err(rstd.cpu().numpy(), r_ref.cpu().numpy()) < 1e-5
    g = y.float().cpu().reshape(B, H * W, 32, 4).permute(0, 2, 1, 3).reshape(B * 32, -1).double()
    assert rel_err(mean.cpu().numpy(), g.mean(1).numpy()) < 1e-5
    assert rel_err(rstd.cpu().numpy(), (1.0 / torch.sqrt(g.var(1, unbiased=False) + 1e-6)).numpy()) < 1e-5
    # images narrower than the statistics kernel's 16-pixel stride (the pixel walk wraps more than once per trip)
    for (hn, wn) in ((9, 5), (3, 16), (40, 7)):
        xn = t(synth.normal(12, (B, hn, wn)))
        yn, (mn, rn) = ops.conv_in_c1(xn.to(DEV), w.to(DEV), b.to(DEV), DT[dt], stats_eps=1e-6)
        refn = F.conv2d(xn.to(DT[dt]).float()[:, None], w, b, padding=1).permute(0, 2, 3, 1)
        assert rel_err(yn.float().cpu().numpy(), refn.numpy()) < (1e-6 if dt == "f32" else 8e-3), (hn, wn)
        gn = yn.float().cpu().reshape(B, hn * wn, 32, 4).permute(0, 2, 1, 3).reshape(B * 32, -1).double()
        assert rel_err(mn.cpu().numpy(), gn.mean(1).numpy()) < 1e-5, (hn, wn)
        assert rel_err(rn.cpu().numpy(), (1.0 / torch.sqrt(gn.var(1, unbiased=False) + 1e-6)).numpy()) < 1e-5, (hn, wn)
    # an image too wide for the statistics kernel's LDS window (2 048 + 2 W + 2 pixels): the plain stem serves, no statistics
    xw = t(synth.normal(13, (1, 3, 6000)))
    yw, none_w = ops.conv_in_c1(xw.to(DEV), w.to(DEV), b.to(DEV), DT[dt], stats_eps=1e-6)
    assert none_w is None
    refw = F.conv2d(xw.to(DT[dt]).float()[:, None], w, b, padding=1).permute(0, 2, 3, 1)
    assert rel_err(yw.float().cpu().numpy(), refw.numpy()) < (1e-6 if dt == "f32" else 8e-3)
    # 128 -> 1
    h = t(synth.normal(8, (B, 20, 53, C))).to(DT[dt])
    wo = t(synth.uniform(9, (1, C, 3, 3), -0.05, 0.05))
    bo = t(synth.uniform(10, (1,), -0.1, 0.1))
    wt = ops.repack_conv_weight(wo.to(DEV), torch.float32).reshape(9, C)
    yo = ops.conv_out_c1(h.to(DEV), wt, bo.to(DEV))
    refo = F.conv2d(h.float().permute(0, 3, 1, 2), wo, bo, padding=1)[:, 0]
    assert rel_err(yo.cpu().numpy(), refo.numpy()) < 2e-5
    # NCHW <-> NHWC boundary copies
    z = t(synth.normal(11, (2, 256, 5, 53)))
    zn = ops.to_nhwc(z.to(DEV), DT[dt])
    assert torch.equal(zn.cpu(), z.permute(0, 2, 3, 1).contiguous().to(DT[dt]))
    back = ops.to_nchw_contiguous(zn, torch.float32)
    assert torch.equal(back.cpu(), z.to(DT[dt]).float())


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_attn_block_vs_oracle(dt):
    from melspec_gpt_vqvae_amd.vqvae.big_model_attn_gan import AttnBlock, set_compute_dtype
    from oracle import vqvae as ovq

    C = 512
    sd = {}
    synth._attn(sd, "a", C, 77)
    sd = {k[2:]: v for k, v in sd.items()}
    blk = _load(AttnBlock(C), sd).to(DEV)
    set_compute_dtype(blk, DT[dt])
    x = t(synth.normal(78, (2, C, 5, 53)))
    y = blk(x.to(DEV))
    assert y.shape == (2, C, 5, 53)
    ref = ovq.attn_block({("a." + k): t(v) for k, v in sd.items()}, "a", x)
    assert rel_err(y.detach().float().cpu().numpy(), ref.numpy()) < (1e-5 if dt == "f32" else 2e-2)   # (as the reference's: requires grad)


def test_narrow_encoder_decoder_vs_reference_golden():
    from melspec_gpt_vqvae_amd.vqvae.big_model_attn_gan import Decoder, Encoder

    g = golden("vqvae_narrow")
    hp = dict(ch=32, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, z_channels=64)
    kw = dict(ch=32, out_ch=1, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, attn_resolutions=[53], in_channels=1,
              resolution=848, z_channels=64, double_z=False)
    enc = _load(Encoder(**kw), synth.encoder_state_dict(int(g["seed"]), **hp)).to(DEV)
    dec = _load(Decoder(**kw), synth.decoder_state_dict(int(g["seed"]), **hp)).to(DEV)
    with torch.no_grad():
        h = enc(t(g["x"], DEV))
        y = dec(t(g["dec_in"], DEV))
    assert h.shape == (1, 64, 5, 53) and y.shape == (1, 1, 80, 848)
    assert rel_err(h.float().cpu().numpy(), g["enc_out"]) < 1e-4
    assert rel_err(y.float().cpu().numpy(), g["dec_out"]) < 1e-4


def test_full_vqvae_tile_to_codes_vs_reference_golden():
    """full-size LitVQVAE (ch=128): mel tile -> encode -> VQ -> codes, and decode of the quantised latent."""
    from melspec_gpt_vqvae_amd.vqvae.big_model_attn_gan import LitVQVAE

    g = golden("vqvae_full")
    m = LitVQVAE(num_embeddings=128, embedding_dim=256)
    assert [k for k in m.state_dict().keys()] == [str(k) for k in g["sd_keys"]], "checkpoint ABI: key names and order"
    _load(m, synth.vqvae_state_dict(int(g["seed"])), allow_missing_prefix="discriminator.")
    m.to(DEV).eval()
    x = t(g["x"], DEV)
    with torch.no_grad():
        z = m.encode(x)
        assert z.shape == (2, 256, 5, 53)
        assert rel_err(z.float().cpu().numpy(), g["z"]) < 1e-4
        codes = m.encode_to_codes(x)
        assert codes.shape == (2, 5, 53) and codes.dtype == torch.int64
        # the encoder's own 1e-6-level differences can move a distance by a few hundred ulp: ties within 2048 ulp
        # may resolve to either of the two nearest codes, everything else must be bit-identical
        n_near, n_flip = check_indices_with_tie_policy(codes.cpu().numpy(), g["indices"], g["gap_ulps"], g["top2"],
                                                       ulp_thresh=2048.0, counts=True)
        listed = int((g["gap_ulps"] < 2048.0).sum())     # 4 of 530 in the fixture; none below 64 ulp
        report("vqvae_full_f32_codes_vs_reference", vectors=530, near_ties_lt_2048ulp=n_near,
               resolved_to_other_code=n_flip, fixture_listed=listed, fixture_lt_8ulp=int((g["gap_ulps"] < 8.0).sum()))
        assert n_near == listed <= 4 and n_flip <= n_near
        loss, q, (perp, enc1h, idx) = m._vq_vae(z)
        assert abs(loss.item() - float(g["vq_loss"])) <= 1e-4 * abs(float(g["vq_loss"]))
        assert abs(perp.item() - float(g["perplexity"])) <= 1e-3 * float(g["perplexity"])
        rec = m.decode(q[:1])
    assert rec.shape == (1, 1, 80, 848)
    assert rel_err(rec.float().cpu().numpy(), g["rec"]) < 1e-4


def test_oversized_inference_batches_are_split_and_give_the_same_results(monkeypatch):
    """encode_to_codes / decode split a batch whose full-resolution activation would pass the kernels' 4 GiB buffer range
    (LitVQVAE._chunks): with the budget lowered to one tile per chunk, three tiles give the codes and the reconstruction
    of the unsplit call bit for bit (GroupNorm statistics are per image, nothing couples the tiles)."""
    from melspec_gpt_vqvae_amd.vqvae.big_model_attn_gan import LitVQVAE

    assert LitVQVAE._chunks(128, 80 * 848) is None and LitVQVAE._chunks(256, 80 * 848) == [(0, 128), (128, 256)]
    assert LitVQVAE._chunks(300, 256 * 5 * 53) == [(0, 128), (128, 256), (256, 300)]
    g = golden("vqvae_full")
    m = LitVQVAE(num_embeddings=128, embedding_dim=256)
    _load(m, synth.vqvae_state_dict(int(g["seed"])), allow_missing_prefix="discriminator.")
    m.to(DEV).eval()
    x = t(g["x"], DEV)
    x3 = torch.cat([x, x[:1].flip(-1)], 0)
    with torch.no_grad():
        codes = m.encode_to_codes(x3)
        q = m._vq_vae.get_codebook_entry(codes.reshape(-1), shape=(3, 5, 53, 256))
        rec = m.decode(q)
        monkeypatch.setattr(LitVQVAE, "_CHUNK_PIXELS", 80 * 848)
        assert LitVQVAE._chunks(3, 80 * 848) == [(0, 1), (1, 2), (2, 3)]
        assert torch.equal(m.encode_to_codes(x3), codes)
        assert torch.equal(m.decode(q), rec)


def _flip_margins(g):
    """Per fixture vector, from the f32 reference's latent (f64 arithmetic): the distances to all 128 codes and, for
    any code c, the FLIP MARGIN rho(c) = (d_c - d_min) / (2 |e_c - e_min|) - how far the latent has to move along the
    unit direction e_c - e_min before c becomes the nearest code (|z| <= 1.0 in this fixture).  A lane whose latent
    error has projection sigma on a fixed direction re-decides vectors with rho of a few sigma and no others."""
    sd = synth.vqvae_state_dict(int(g["seed"]))
    E = sd["_vq_vae._embedding.weight"].astype(np.float64)
    z = g["z"].astype(np.float64).transpose(0, 2, 3, 1).reshape(-1, E.shape[1])
    d = (z ** 2).sum(1)[:, None] + (E ** 2).sum(1)[None] - 2.0 * z @ E.T
    best = d.argmin(1)
    assert np.array_equal(best, g["indices"].astype(np.int64))
    n = np.arange(len(z))
    en = np.linalg.norm(E[:, None, :] - E[None, :, :], axis=-1)          # (128, 128) code-to-code distances
    rho = (d - d[n, best][:, None]) / (2.0 * np.maximum(en[best], 1e-30))  # (530, 128); rho[n, best[n]] = 0
    return z, E, best, rho


# Resolution of the 16-bit lanes on this fixture, fixed here from profiles/r04_parity_report.jsonl: the bf16 encoder's
# latent error projected on a vector's (nearest, second-nearest) code direction has an rms of BF16_PROJ_RMS; the
# disagreements of all three lookup paths sit at flip margins below BF16_FLIP_RES (= a few rms).  A change that only
# re-decides near-ties moves WHICH of the ~40 vectors inside the resolution flip; a real regression either raises the
# projected error or flips a vector outside it - both fail below, neither depends on the count of coin flips.
BF16_FLIP_RES = 0.012
BF16_PROJ_RMS = 4.0e-3


def test_full_vqvae_bf16_lane_reports_code_agreement():
    """bf16 throughput lane of the encoder against the f32 reference (north_star's "indices bit-exact" is the f32 lane's
    property; the 16-bit lane is held to its own resolution): every code that differs from the reference's is on a vector
    whose reference flip margin is inside the lane's resolution, none outside; latent error bounded elementwise and in
    projection on the deciding directions."""
    from melspec_gpt_vqvae_amd.vqvae.big_model_attn_gan import LitVQVAE, set_compute_dtype

    g = golden("vqvae_full")
    m = LitVQVAE(num_embeddings=128, embedding_dim=256)
    _load(m, synth.vqvae_state_dict(int(g["seed"])), allow_missing_prefix="discriminator.")
    m.to(DEV).eval()
    set_compute_dtype(m, torch.bfloat16)
    with torch.no_grad():
        z = m.encode(t(g["x"], DEV))
        codes = m.encode_to_codes(t(g["x"], DEV))                       # quant_conv folded into the codebook image
        codes_unfused = m.encode_to_codes(t(g["x"], DEV), fused=False)  # quant_conv -> bf16 z -> lookup
        codes_lo = m._vq_vae.encode_indices_fused(m._encoder._nhwc(t(g["x"], DEV)).permute(0, 3, 1, 2), m.quant_conv,
                                                  with_lo=True)
    err = rel_err(z.float().cpu().numpy(), g["z"])
    zref, E, best, rho = _flip_margins(g)
    nvec = len(best)
    n = np.arange(nvec)
    # the lane's latent error along each vector's deciding direction (nearest -> second-nearest code)
    second = np.where(np.arange(128)[None] == best[:, None], np.inf, rho).argmin(1)
    dz = z.float().cpu().numpy().astype(np.float64).transpose(0, 2, 3, 1).reshape(nvec, -1) - zref
    u = E[second] - E[best]
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    proj = (dz * u).sum(1)
    proj_rms, proj_max = float(np.sqrt((proj ** 2).mean())), float(np.abs(proj).max())
    at_risk = int((rho[n, second] <= BF16_FLIP_RES).sum())
    rec = dict(latent_rel_to_max_err=err, latent_proj_rms=proj_rms, latent_proj_max=proj_max, vectors=nvec,
               flip_resolution=BF16_FLIP_RES, vectors_inside_resolution=at_risk)
    worst = 0.0
    for name, c in (("fused_hi_only", codes), ("unfused", codes_unfused), ("fused_hi_lo", codes_lo)):
        got = c.cpu().numpy().ravel().astype(np.int64)
        diff = np.nonzero(got != best)[0]
        margins = rho[diff, got[diff]]
        rec["code_agreement_" + name] = float((got == best).mean())
        rec["flips_" + name] = int(diff.size)
        rec["max_flip_margin_" + name] = float(margins.max()) if diff.size else 0.0
        rec["flips_outside_resolution_" + name] = int((margins > BF16_FLIP_RES).sum())
        worst = max(worst, rec["max_flip_margin_" + name])
    report("vqvae_full_bf16_lane_vs_f32_reference", **rec)
    assert err < 5e-2 and proj_rms <= 1.5 * BF16_PROJ_RMS and proj_max <= 6 * 1.5 * BF16_PROJ_RMS, rec
    for name in ("fused_hi_only", "unfused", "fused_hi_lo"):
        assert rec["flips_outside_resolution_" + name] == 0, (name, rec)
        assert rec["flips_" + name] <= at_risk, (name, rec)        # (cannot exceed the population it is drawn from)
        assert rec["code_agreement_" + name] >= 0.95, (name, rec)  # sanity floor only; the gates are the two above


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("H,W,Cin,Cout,B", [(80, 848, 128, 128, 1), (20, 53, 128, 256, 2), (5, 7, 256, 128, 3),
                                             (9, 33, 128, 64, 2)])
def test_fused_groupnorm_swish_conv3x3(dt, H, W, Cin, Cout, B):
    """halo-tiled conv with GroupNorm+swish applied while staging (csrc/conv_fused.hip) == GN -> swish -> conv2d."""
    from melspec_gpt_vqvae_amd import ops

    from melspec_gpt_vqvae_amd import _ffi

    x = t(synth.normal(1, (B, H, W, Cin), 1.3, 0.5)).to(DT[dt])
    w = t(synth.normal(2, (Cout, Cin, 3, 3), 0.03)).to(DT[dt])
    bias = t(synth.normal(3, (Cout,), 0.1))
    gm, bt = t(synth.normal(4, (Cin,), 0.1, 1.0)), t(synth.normal(5, (Cin,), 0.1))
    res = t(synth.normal(6, (B, H, W, Cout))).to(DT[dt])
    xd = x.to(DEV)
    stats = ops.groupnorm_stats(xd, 1e-6)
    wp = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    if not ops.fused_conv_supported(Cin, DT[dt]):
        # the patch does not fit LDS for this width / dtype (f32, 256 channels): the entry point refuses LOUDLY, nothing is
        # launched (the module takes GroupNorm + the implicit-GEMM conv there: tests of the full encoder / decoder)
        with pytest.raises(_ffi.MelgptError):
            ops.conv3x3_gn(xd, stats, gm.to(DEV), bt.to(DEV), wp, bias.to(DEV), swish=True, residual=res.to(DEV))
        return
    y = ops.conv3x3_gn(xd, stats, gm.to(DEV), bt.to(DEV), wp, bias.to(DEV), swish=True, residual=res.to(DEV))
    h = F.group_norm(x.float().permute(0, 3, 1, 2), 32, gm, bt, eps=1e-6)
    h = h * torch.sigmoid(h)
    if dt == "bf16":
        h = h.to(torch.bfloat16).float()   # the kernel rounds the normalised patch to bf16 before the MFMAs
    ref = F.conv2d(h, w.float(), bias, padding=1).permute(0, 2, 3, 1) + res.float()
    assert rel_err(y.float().cpu().numpy(), ref.numpy()) < (2e-5 if dt == "f32" else 8e-3)
    # plain (no norm) mode
    y2 = ops.conv3x3_gn(xd, None, None, None, wp, bias.to(DEV), swish=False)
    ref2 = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias, padding=1).permute(0, 2, 3, 1)
    assert rel_err(y2.float().cpu().numpy(), ref2.numpy()) < (2e-5 if dt == "f32" else 8e-3)


@pytest.mark.parametrize("with_res", [False, True])
@pytest.mark.parametrize("H,W,B", [(80, 848, 8), (40, 424, 9), (37, 250, 16)])
def test_fused_conv_also_yields_groupnorm_stats_of_its_output(H, W, B, with_res):
    """bf16, 128 -> 128, with and without the shortcut operand: the persistent fused conv accumulates the GroupNorm(32)
    statistics of its own output in its epilogue (melgpt_conv3x3_gn_nhwc_stats).  Same tensor as the plain call,
    statistics equal to a separate melgpt_groupnorm_stats pass over it (edge tiles and padded rows included)."""
    from melspec_gpt_vqvae_amd import ops

    C = 128
    x = t(synth.normal(11, (B, H, W, C), 1.1, 0.3)).to(torch.bfloat16).to(DEV)
    w = t(synth.normal(12, (C, C, 3, 3), 0.03)).to(torch.bfloat16)
    bias = t(synth.normal(13, (C,), 0.2)).to(DEV)
    gm, bt = t(synth.normal(14, (C,), 0.1, 1.0)).to(DEV), t(synth.normal(15, (C,), 0.1)).to(DEV)
    wp = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    stats = ops.groupnorm_stats(x, 1e-6)
    res = t(synth.normal(16, (B, H, W, C), 0.7)).to(torch.bfloat16).to(DEV) if with_res else None
    r = ops.conv3x3_gn_with_out_stats(x, stats, gm, bt, wp, bias, 1e-6, swish=True, residual=res)
    y0 = ops.conv3x3_gn(x, stats, gm, bt, wp, bias, swish=True, residual=res)
    if r is None:
        pytest.skip("this shape does not run on the persistent fused kernel")
    y, (mean, rstd) = r
    assert torch.equal(y, y0)
    # the persistent kernel against torch's own GroupNorm -> swish -> conv2d in f32 (16 x 16-pixel tiles at 80 x 848 and
    # 37 x 250, 8 x 32 at 40 x 424; edge tiles in both)
    h = F.group_norm(x.float().permute(0, 3, 1, 2), 32, gm, bt, eps=1e-6)
    h = (h * torch.sigmoid(h)).to(torch.bfloat16).float()
    ref = F.conv2d(h, w.float().to(DEV), bias, padding=1).permute(0, 2, 3, 1)
    if with_res:
        ref = ref + res.float()
    assert rel_err(y.float().cpu().numpy(), ref.cpu().numpy()) < 8e-3
    m2, r2 = ops.groupnorm_stats(y, 1e-6)
    assert float((mean - m2).abs().max()) < 2e-5 * max(1.0, float(m2.abs().max()))
    assert float(((rstd - r2) / r2).abs().max()) < 1e-4
    assert ops.conv3x3_gn_with_out_stats(x.float(), None, None, None, wp.float(), bias, 1e-6, swish=False) is None
    # the same launch geometry without a norm in front (plain 3 x 3 convolution of the raw activation), and with the norm
    # but without swish: both staging branches of the wave-specialised kernel
    y3 = ops.conv3x3_gn(x, None, None, None, wp, bias, swish=False, residual=res)
    ref3 = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().to(DEV), bias, padding=1).permute(0, 2, 3, 1)
    if with_res:
        ref3 = ref3 + res.float()
    assert rel_err(y3.float().cpu().numpy(), ref3.cpu().numpy()) < 8e-3
    y4 = ops.conv3x3_gn(x, stats, gm, bt, wp, bias, swish=False)
    h4 = F.group_norm(x.float().permute(0, 3, 1, 2), 32, gm, bt, eps=1e-6).to(torch.bfloat16).float()
    ref4 = F.conv2d(h4, w.float().to(DEV), bias, padding=1).permute(0, 2, 3, 1)
    assert rel_err(y4.float().cpu().numpy(), ref4.cpu().numpy()) < 8e-3


def test_extract_codes_writes_reference_named_files(tmp_path):
    """feature_extraction/extract_codes.py (:31-56): <class>/melspec_10s_22050hz/<v>_mel.npy -> <class>/codes_10s/
    <v>_mel_code.npy, (5, 53) int64, equal to encoding the centre crop 2x-1 directly; existing files are skipped."""
    import os

    from melspec_gpt_vqvae_amd.feature_extraction.extract_codes import extract_all
    from melspec_gpt_vqvae_amd.vqvae.big_model_attn_gan import LitVQVAE

    g = golden("vqvae_full")
    m = LitVQVAE(num_embeddings=128, embedding_dim=256)
    _load(m, synth.vqvae_state_dict(int(g["seed"])), allow_missing_prefix="discriminator.")
    m.to(DEV).eval()
    root = os.path.join(str(tmp_path), "vas", "features")
    mels = synth.mel_tiles(21, 3)                       # (3, 80, 860) in [0, 1]
    for i, cls in enumerate(["dog", "dog", "gun"]):
        d = os.path.join(root, cls, "melspec_10s_22050hz")
        os.makedirs(d, exist_ok=True)
        np.save(os.path.join(d, f"video_{i:05d}_mel.npy"), mels[i])
    written = extract_all(root, m, DEV, 848, batch_size=2)
    assert sorted(os.path.relpath(w, root) for w in written) == [
        "dog/codes_10s/video_00000_mel_code.npy", "dog/codes_10s/video_00001_mel_code.npy",
        "gun/codes_10s/video_00002_mel_code.npy"]
    with torch.no_grad():
        ref = m.encode_to_codes(t(2 * mels[:, :, 6:854] - 1, DEV).unsqueeze(1)).cpu().numpy()
    for i, cls in enumerate(["dog", "dog", "gun"]):
        c = np.load(os.path.join(root, cls, "codes_10s", f"video_{i:05d}_mel_code.npy"))
        assert c.shape == (5, 53) and c.dtype == np.int64 and np.array_equal(c, ref[i])
    assert extract_all(root, m, DEV, 848) == []         # nothing left to do


def test_backward_through_the_full_size_vqvae_runs_and_frozen_modules_record_nothing():
    """The reference's LitVQVAE.forward is differentiable end to end (big_model_attn_gan.py:622-634); since round 6 so is this
    one (vqvae/autograd.py; parity of every gradient on the narrow model: tests/test_vqvae_bwd_gpu.py).  Here the FULL-SIZE
    model, one tile, f32 lane: forward + backward run, every encoder / decoder / codebook parameter receives a finite gradient;
    a gradient w.r.t. the decoder's latent input comes back in its layout; a frozen module fed a plain input records nothing
    and the quantiser alone stays differentiable."""
    from melspec_gpt_vqvae_amd.vqvae.big_model_attn_gan import LitVQVAE

    g = golden("vqvae_full")
    m = LitVQVAE(num_embeddings=128, embedding_dim=256)
    _load(m, synth.vqvae_state_dict(int(g["seed"])), allow_missing_prefix="discriminator.")
    m.to(DEV).train()
    x = t(g["x"][:1], DEV)
    with torch.no_grad():
        z0 = m.encode(x)
    z = m.encode(x)                                  # autograd on, parameters require a gradient: the differentiable path
    assert z.requires_grad and rel_err(z.detach().cpu().numpy(), z0.cpu().numpy()) < 1e-4
    loss, x_recon, info = m(x)
    assert x_recon.shape == x.shape and x_recon.requires_grad and loss.requires_grad
    (loss + x_recon.float().mean()).backward()
    for n, p in m.named_parameters():
        if n.startswith("discriminator."):
            assert p.grad is None
        else:
            assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.shape == p.shape, n
    zq = z0.clone().requires_grad_(True)
    m.decode(zq).float().sum().backward()
    assert zq.grad is not None and zq.grad.shape == zq.shape and torch.isfinite(zq.grad.float()).all()
    # a frozen module fed a plain input records nothing, and the quantiser alone stays differentiable (its own kernel)
    for p in m.parameters():
        p.requires_grad_(False)
    assert not m.encode(x).requires_grad
    zl = z0.clone().requires_grad_(True)
    l2, q, _ = m._vq_vae(zl)
    (l2 + q.float().sum()).backward()
    assert zl.grad is not None and torch.isfinite(zl.grad.float()).all()


def test_backward_through_the_vqvae_in_the_16_bit_lane_points_the_same_way_as_the_f32_lane():
    """the differentiable path in the 16-bit lane (16-bit activations and gradients between layers, f32 accumulation, f32
    parameter gradients), full-size model, one tile: every gradient finite, and the gradients of a sample of parameters along the
    whole depth within a cosine of 0.97 of the f32 lane's (decoder only: behind the quantiser a flipped code would change the target)."""
    from melspec_gpt_vqvae_amd.vqvae.big_model_attn_gan import LitVQVAE, set_compute_dtype

    g = golden("vqvae_full")
    grads = {}
    zq = None
    for dt in (torch.float32, torch.bfloat16):
        m = LitVQVAE(num_embeddings=128, embedding_dim=256)
        _load(m, synth.vqvae_state_dict(int(g["seed"])), allow_missing_prefix="discriminator.")
        m.to(DEV).train()
        set_compute_dtype(m, dt)
        if zq is None:
            with torch.no_grad():
                zq = m._vq_vae(m.encode(t(g["x"][:1], DEV)))[1].detach()          # one quantised latent for both lanes
        rec = m.decode(zq.to(dt) if dt != torch.float32 else zq)
        w = t(synth.normal(77, tuple(rec.shape), 1.0), DEV)
        (rec.float() * w).sum().backward()
        grads[dt] = {n: p.grad.detach().float().flatten() for n, p in m.named_parameters() if p.grad is not None}
        assert all(torch.isfinite(v).all() for v in grads[dt].values())
    names = ["post_quant_conv.weight", "_decoder.conv_in.weight", "_decoder.mid.attn_1.q.weight", "_decoder.up.4.block.0.conv1.weight",
             "_decoder.up.3.upsample.conv.weight", "_decoder.up.1.block.0.nin_shortcut.weight", "_decoder.up.0.block.2.conv2.weight",
             "_decoder.norm_out.weight", "_decoder.conv_out.weight"]
    cos = {n: float(torch.nn.functional.cosine_similarity(grads[torch.float32][n], grads[torch.bfloat16][n], dim=0)) for n in names}
    report("vqvae_decoder_grad_cosine_bf16_vs_f32", **cos)
    assert all(c > 0.97 for c in cos.values()), cos
