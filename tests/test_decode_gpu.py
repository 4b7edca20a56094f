"""KV-cached decoding (GPT.decode_step / csrc/decode.hip) against the full re-forward the reference's
sampling loops run per token (transformer/minGPT.py:293-360, decoders.py:89-123), and against the greedy samples
recorded from the real reference (tests/golden/lit_mingpt.npz).  f32 lane gate 1e-4 on logits; samples bit-exact."""
import numpy as np
import pytest
import torch

import synth
from util import golden, rel_err, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _load(module, sd_np):
    res = module.load_state_dict({k: t(v) for k, v in sd_np.items()}, strict=False)
    assert not res.unexpected_keys and all(k.endswith('attn.mask') for k in res.missing_keys), res
    return module


def _lit():
    from melspec_gpt_vqvae_amd.transformer.minGPT import Lit_minGPT

    g = golden("lit_mingpt")
    args = synth.gpt_args(n_layer=2, n_head=4, n_embd=256, reconstruct_spec="", device=DEV, batch_size=2, learning_rate=1e-6)
    lit = Lit_minGPT(args)
    _load(lit.transformer, synth.gpt_state_dict(args, int(g["sd_seed"])))
    lit.to(DEV).eval()
    batch = {"codes": t(g["codes"], DEV), "target": t(g["target"], DEV)}
    return lit, g, lit.get_x(batch), lit.get_c(batch)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_decode_step_logits_equal_last_row_of_full_forward(dt):
    from melspec_gpt_vqvae_amd.transformer.minGPT import set_compute_dtype

    lit, g, x, c = _lit()
    tr = lit.transformer
    if dt == "bf16":
        set_compute_dtype(tr, torch.bfloat16)
    # bf16 lane: the two paths round their intermediates at different places (reported, not gated at 1e-4)
    tol = 1e-4 if dt == "f32" else 8e-2
    with torch.no_grad():
        cache = tr.decode_begin(x.size(0))
        logits = tr.decode_step(cache, pre_idx=c)
        full, _, _ = tr(x[:, :0], c)
        assert rel_err(logits.cpu().numpy(), full[:, -1].cpu().numpy()) < tol
        for j in range(40):
            want_att = j == 39
            out = tr.decode_step(cache, idx=x[:, j:j + 1], want_att=want_att)
            logits, att_row = out if want_att else (out, None)
            if j in (0, 1, 7, 39):
                full, _, att = tr(x[:, :j + 1], c)
                assert rel_err(logits.cpu().numpy(), full[:, -1].cpu().numpy()) < tol, j
        # attention row of the last block for the newest position = last row of the full map
        T = 41
        assert rel_err(att_row[:, :, :T].cpu().numpy(), att[:, :, -1, :].cpu().numpy()) < (1e-5 if dt == "f32" else 2e-2)
    assert cache["pos"] == 41


def test_cached_and_reforward_sampling_match_reference_golden():
    lit, g, x, c = _lit()
    for kv in (True, False):
        xs, att = lit.sample(x[:, :9], c, steps=16, sample=False, kv_cache=kv)
        assert np.array_equal(xs.cpu().numpy(), g["greedy16"]), kv
        assert list(att.shape) == list(g["att_shape"]) and not att.is_cuda
        assert rel_err(att.numpy()[:, :, -1], g["att_last"]) < 1e-4
        xk, _ = lit.sample(x[:, :9], c, steps=4, sample=False, top_k=5, temperature=0.7, kv_cache=kv)
        assert np.array_equal(xk.cpu().numpy(), g["greedy4_topk"]), kv
    # stochastic sampling: same seed stream => same draws on both paths (the draw depends on (seed, step) only)
    from melspec_gpt_vqvae_amd.transformer import minGPT as mg
    outs = []
    for kv in (True, False):
        mg._Seeds.counter = 1000   # sample() draws its seed first thing on both paths
        outs.append(lit.sample(x[:, :3], c, steps=12, sample=True, top_k=20, kv_cache=kv)[0].cpu().numpy())
    assert np.array_equal(outs[0], outs[1])


def test_full_length_generation_from_class_token_only():
    """265 tokens from the class token alone, VAS-size block (266 positions): cache fills to the last slot."""
    lit, g, x, c = _lit()
    empty = x[:, :0]
    xs, att = lit.sample(empty, c, steps=265, sample=False)
    assert xs.shape == (2, 265) and int(xs.min()) >= 0 and int(xs.max()) < 128
    ref, _ = lit.sample(empty, c, steps=24, sample=False, kv_cache=False)
    assert np.array_equal(xs[:, :24].cpu().numpy(), ref.cpu().numpy())
    assert att.shape[-1] == 265


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(1, 1024, 1024), (3, 384, 256), (16, 128, 1024), (37, 512, 2048), (128, 256, 512)])
def test_linear_rows_weight_streaming_kernel(dt, M, N, K):
    """melgpt_gemv_rows == x @ W^T + b -> exact-erf GELU / + residual (nn.Linear, nn.GELU: minGPT.py:100-104)."""
    import torch.nn.functional as F

    from melspec_gpt_vqvae_amd import ops

    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    x = t(synth.normal(1, (M, K), 1.0)).to(tdt)
    w = t(synth.normal(2, (N, K), 0.05)).to(tdt)
    b = t(synth.normal(3, (N,), 0.1))
    r = t(synth.normal(4, (M, N), 1.0)).to(tdt)
    ref = x.float() @ w.float().T + b
    tol = 2e-5 if dt == "f32" else 1e-2
    y = ops.linear_rows(x.to(DEV), w.to(DEV), bias=b.to(DEV))
    assert y.dtype == tdt and rel_err(y.float().cpu().numpy(), ref.numpy()) < tol
    y = ops.linear_rows(x.to(DEV), w.to(DEV), bias=b.to(DEV), act=ops.ACT_GELU)
    assert rel_err(y.float().cpu().numpy(), F.gelu(ref).numpy()) < tol
    y = ops.linear_rows(x.to(DEV), w.to(DEV), residual=r.to(DEV), out_dtype=torch.float32)
    assert y.dtype == torch.float32
    assert rel_err(y.cpu().numpy(), (x.float() @ w.float().T + r.float()).numpy()) < tol
    # LayerNorm folded into the layer (pre-LN of a transformer block): y = LN(x) W^T + b
    gm, bt = t(synth.normal(5, (K,), 0.1, 1.0)), t(synth.normal(6, (K,), 0.1))
    h = F.layer_norm(x.float() * 3 + 0.5, (K,), gm, bt, 1e-5)
    if dt == "bf16":
        h = h.to(torch.bfloat16).float()          # the kernel rounds LN(x) to bf16 like the separate LayerNorm kernel
    xs = (x.float() * 3 + 0.5).to(tdt)
    if dt == "bf16":
        h = F.layer_norm(xs.float(), (K,), gm, bt, 1e-5).to(torch.bfloat16).float()
    y = ops.linear_rows(xs.to(DEV), w.to(DEV), bias=b.to(DEV), ln=(gm.to(DEV), bt.to(DEV), 1e-5), out_dtype=torch.float32)
    # (bf16, 5+ rows, K = 1024: the LN is folded into the weight algebraically - W'x - mu c1 cancels the row mean in f32
    # after the product instead of before it; same bound as test_skinny_linear_with_layernorm_folded_in)
    assert rel_err(y.cpu().numpy(), (h @ w.float().T + b).numpy()) < (2e-5 if dt == "f32" else 8e-3)


@pytest.mark.parametrize("M", [5, 16, 17, 33, 64, 100, 128])
@pytest.mark.parametrize("N,K,act,res,f32out", [(1024, 1024, 0, True, False), (4096, 1024, 1, False, False),
                                                (1024, 4096, 0, True, False), (128, 1024, 0, False, True)])
def test_skinny_mfma_linear_matches_torch(M, N, K, act, res, f32out):
    """5..128 bf16 rows go through melgpt_linear_skinny (one workgroup per 16 output columns, four waves split K);
    same rounding points as the tiled GEMM: f32 accumulation, bias + exact-erf GELU + residual in f32, one rounding."""
    from melspec_gpt_vqvae_amd import ops

    x = (synth.normal(300 + M, (M, K)) * 0.5).astype(np.float32)
    w = (synth.normal(301, (N, K)) * 0.05).astype(np.float32)
    b = synth.normal(302, (N,)).astype(np.float32)
    r = synth.normal(303, (M, N)).astype(np.float32)
    xb, wb, rb = (t(a, DEV).to(torch.bfloat16) for a in (x, w, r))
    y = ops.linear_rows(xb, wb, bias=t(b, DEV), act=act, residual=rb if res else None,
                        out_dtype=torch.float32 if f32out else None)
    ref = xb.float().cpu() @ wb.float().cpu().T + t(b)
    if act:
        ref = torch.nn.functional.gelu(ref)
    if res:
        ref = ref + rb.float().cpu()
    assert y.dtype == (torch.float32 if f32out else torch.bfloat16) and y.shape == (M, N)
    assert rel_err(y.float().cpu().numpy(), ref.numpy()) < (2e-5 if f32out else 6e-3)
    y2 = ops.gemm(xb, wb, bias=t(b, DEV), act=act, residual=rb if res else None, out_dtype=torch.float32 if f32out else None)
    assert rel_err(y.float().cpu().numpy(), y2.float().cpu().numpy()) < (2e-5 if f32out else 6e-3)


@pytest.mark.parametrize("M", [5, 16, 31, 64])
@pytest.mark.parametrize("N,K,act", [(3072, 1024, 0), (4096, 1024, 1), (256, 512, 0)])
def test_skinny_linear_with_layernorm_folded_in(M, N, K, act):
    """y = W LN(x) in one launch (the pre-LN of a Block folded into its qkv / fc1 layer at batch 5 .. 64): row statistics
    from the MFMA operand registers; same rounding points as LayerNorm kernel -> linear (normalised rows rounded to
    bf16), so the two agree to bf16 rounding of a few borderline elements."""
    from melspec_gpt_vqvae_amd import ops

    x = (synth.normal(400 + M, (M, K)) * 1.5 + 0.3).astype(np.float32)
    w = (synth.normal(401, (N, K)) * 0.05).astype(np.float32)
    b = synth.normal(402, (N,)).astype(np.float32)
    gam = (1.0 + 0.2 * synth.normal(403, (K,))).astype(np.float32)
    bet = (0.1 * synth.normal(404, (K,))).astype(np.float32)
    xb, wb = t(x, DEV).to(torch.bfloat16), t(w, DEV).to(torch.bfloat16)
    y = ops.linear_rows(xb, wb, bias=t(b, DEV), act=act, ln=(t(gam, DEV), t(bet, DEV), 1e-5))
    xn = torch.nn.functional.layer_norm(xb.float().cpu(), (K,), t(gam), t(bet), 1e-5).to(torch.bfloat16).float()
    ref = xn @ wb.float().cpu().T + t(b)
    if act:
        ref = torch.nn.functional.gelu(ref)
    assert y.dtype == torch.bfloat16 and y.shape == (M, N)
    assert rel_err(y.float().cpu().numpy(), ref.numpy()) < 8e-3
    # and against the two-launch path (LayerNorm kernel, then the same linear)
    xs = ops.layernorm_fwd(xb, t(gam, DEV), t(bet, DEV), 1e-5, want_stats=False)[0]
    y2 = ops.linear_rows(xs, wb, bias=t(b, DEV), act=act)
    assert rel_err(y.float().cpu().numpy(), y2.float().cpu().numpy()) < 8e-3


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_decode_never_reads_cache_rows_it_has_not_written(dt):
    """the KV cache comes from torch.empty: rows at and beyond the current position hold anything.  The attention step
    requests clamped rows unconditionally (loads in flight together), so their VALUES must never reach the result -
    a cache pre-filled with NaN has to give the same logits as a zeroed one."""
    from melspec_gpt_vqvae_amd.transformer.minGPT import set_compute_dtype

    lit, g, x, c = _lit()
    tr = lit.transformer
    if dt == "bf16":
        set_compute_dtype(tr, torch.bfloat16)
    outs = []
    with torch.no_grad():
        for fill in (0.0, float("nan")):
            cache = tr.decode_begin(x.size(0))
            for buf in cache["k"] + cache["v"]:
                buf.fill_(fill)
            rows = [tr.decode_step(cache, pre_idx=c)]
            for j in range(6):
                rows.append(tr.decode_step(cache, idx=x[:, j:j + 1]))
            outs.append(torch.stack(rows).cpu().numpy())
    assert np.isfinite(outs[1]).all()
    assert np.array_equal(outs[0], outs[1])


def test_sampling_with_an_unmasked_prefix_does_not_use_a_stale_kv_cache():
    """n_unmasked > 0 (reference minGPT.py:65-69): positions below n_unmasked attend bidirectionally, so their K / V
    change as the sequence grows and a cache of them would be stale.  sample(kv_cache=True) must then take the
    reference's re-forward loop (same samples as kv_cache=False, same as the CPU oracle); decode_begin refuses."""
    from oracle import gpt as ogpt

    from melspec_gpt_vqvae_amd.transformer.minGPT import Lit_minGPT

    args = synth.gpt_args(n_layer=2, n_head=4, n_embd=256, n_unmasked=12, reconstruct_spec="", device=DEV, batch_size=2,
                          learning_rate=1e-6)
    lit = Lit_minGPT(args)
    sd_np = synth.gpt_state_dict(args, 1)
    _load(lit.transformer, sd_np)
    lit.to(DEV).eval()
    assert not lit.transformer.kv_cacheable()
    with pytest.raises(AssertionError, match="n_unmasked"):
        lit.transformer.decode_begin(2)
    x0 = t(synth.randint(77, 0, 128, (2, 5)), DEV)
    c = t(synth.randint(78, 0, 8, (2, 1)), DEV)
    a, _ = lit.sample(x0, c, steps=20, sample=False)                  # default kv_cache=True falls back
    b, _ = lit.sample(x0, c, steps=20, sample=False, kv_cache=False)
    assert torch.equal(a, b)
    ref, _ = ogpt.sample_class_gpt(ogpt.as_torch_sd(sd_np), x0.cpu(), c.cpu(), 20, 2, 4, n_unmasked=12)
    assert np.array_equal(a.cpu().numpy(), ref.numpy())


