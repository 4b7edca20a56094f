"""GPU parity of the persistent GEMM's ping-pong K loop (csrc/gemm8p.hip) at sizes where every workgroup walks several
256 x 256 (192 x 256) tiles: against an fp32 matmul of the same bf16 operands on the CPU (1e-5 relative-to-max before
the output rounding, 2^-8 after it), and BIT FOR BIT against the ring kernel (csrc/gemm256.hip) - both loops add the
same products in the same order.  Every case also checks, through the launch counters, that the loop it means to test
is the one that ran (a silent fall-back to the other kernel would make the comparison empty).
Reference call sites: the Linear layers of transformer/minGPT.py:76-88,100-117 (forward, input and weight gradients)."""
import ctypes

import pytest
import torch

from util import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _launches():
    from melspec_gpt_vqvae_amd import _ffi

    r, p = ctypes.c_longlong(0), ctypes.c_longlong(0)
    assert _ffi.lib().melgpt_gemm_loop_launches(ctypes.byref(r), ctypes.byref(p)) == 0
    return r.value, p.value


def _both(fn):
    """fn() on the ring and on the ping-pong loop -> (ring output, ping-pong output), with the counters checked."""
    from melspec_gpt_vqvae_amd import _ffi

    L = _ffi.lib()
    was = L.melgpt_get_gemm_pingpong()
    try:
        L.melgpt_set_gemm_pingpong(0)
        r0, p0 = _launches()
        ring = fn()
        ring = [x.clone() for x in ring] if isinstance(ring, (list, tuple)) else ring.clone()
        r1, p1 = _launches()
        assert r1 > r0 and p1 == p0, "with the switch off every persistent launch takes the ring"
        L.melgpt_set_gemm_pingpong(1)
        pp = fn()
        pp = [x.clone() for x in pp] if isinstance(pp, (list, tuple)) else pp.clone()
        r2, p2 = _launches()
        assert p2 > p1, "this shape was meant to run on the ping-pong loop"
    finally:
        L.melgpt_set_gemm_pingpong(was)
    torch.cuda.synchronize()
    return ring, pp


def _ops(seed, M, N, K, b_kmajor=False):
    torch.manual_seed(seed)
    a = (torch.randn(M, K) * 0.5).to(torch.bfloat16)
    b = (torch.randn(*((K, N) if b_kmajor else (N, K))) * 0.25).to(torch.bfloat16)
    ref = a.float() @ (b.float() if b_kmajor else b.float().t())
    return a.to(DEV), b.to(DEV), ref


# >= 256 tiles (a full grid: the ping-pong kernel walks XCD blocks), ragged M / N / K edges, both tile heights
# (the height is whatever the launch's cost model picks: N = 1024 -> 192 rows)
@pytest.mark.parametrize("form,M,N,K", [("nt", 9000, 4096, 328), ("nt", 33920, 1024, 264), ("nt", 20000, 1272, 384),
                                        ("nn", 24576, 1024, 320), ("nn", 8500, 4096, 328), ("nt", 8192, 8192, 512),
                                        # GPT-VAE XL widths: K-major B whose N ends inside a 128-column half-tile
                                        ("nn", 33920, 1472, 328), ("nn", 12000, 4416, 264)])
def test_plain_store_matches_reference_and_ring(form, M, N, K):
    from melspec_gpt_vqvae_amd import ops

    a, b, ref = _ops(M + N + K, M, N, K, b_kmajor=form == "nn")
    # (an f32 output of an NT / NN product is a ring-only mode: the bf16 store is what the ping-pong kernel builds)
    ring, pp = _both(lambda: ops.gemm(a, b, b_kmajor=form == "nn"))
    assert torch.equal(ring, pp)
    assert rel_err(pp.float().cpu().numpy(), ref.numpy()) < 2 ** -8


def test_fused_epilogues_match_reference_and_ring():
    import torch.nn.functional as F

    from melspec_gpt_vqvae_amd import ops

    M, N, K = 17000, 4096, 264      # 67 x 16 tiles: five rounds, ragged last tile row and a ragged K tile
    a, b, ref = _ops(5, M, N, K)
    bias = (torch.randn(N) * 0.1)
    res = torch.randn(M, N).to(torch.bfloat16)
    bd, rd = bias.to(DEV), res.to(DEV)
    pre_ref = ref + bias
    # bias only (qkv), bias + residual (eval-mode projection)
    ring, pp = _both(lambda: ops.gemm(a, b, bias=bd))
    assert torch.equal(ring, pp) and rel_err(pp.float().cpu().numpy(), pre_ref.numpy()) < 2 ** -8
    ring, pp = _both(lambda: ops.gemm(a, b, bias=bd, residual=rd))
    assert torch.equal(ring, pp) and rel_err(pp.float().cpu().numpy(), (pre_ref + res.float()).numpy()) < 2 ** -8
    # Linear -> GELU with the saved derivative (fc1 in training)
    def fc1():
        pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        act = ops.gemm(a, b, bias=bd, act=ops.ACT_GELU_DACT, pre_out=pre)
        return [act, pre]
    ring, pp = _both(fc1)
    assert torch.equal(ring[0], pp[0]) and torch.equal(ring[1], pp[1])
    assert rel_err(pp[0].float().cpu().numpy(), F.gelu(pre_ref).numpy()) < 2 ** -8
    x = pre_ref
    gprime = 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
    assert rel_err(pp[1].float().cpu().numpy(), gprime.numpy()) < 2 ** -7
    # Linear -> dropout -> + residual (projection / fc2 in training); the mask is the one dropout_apply replays
    ring, pp = _both(lambda: ops.gemm(a, b, bias=bd, residual=rd, drop_p=0.25, seed=77, stream_id=5))
    assert torch.equal(ring, pp)
    mask = ops.dropout_apply(torch.ones(M, N, dtype=torch.bfloat16, device=DEV), 0.25, 77, 5).float().cpu()
    assert rel_err(pp.float().cpu().numpy(), (pre_ref * mask + res.float()).numpy()) < 2 ** -7
    # input gradient through the saved GELU derivative: (dY W) * R, K-major B
    a2, b2, ref2 = _ops(6, M, N, K, b_kmajor=True)
    ring, pp = _both(lambda: ops.gemm(a2, b2, b_kmajor=True, act=ops.ACT_MUL, residual=rd))
    assert torch.equal(ring, pp) and rel_err(pp.float().cpu().numpy(), (ref2 * res.float()).numpy()) < 2 ** -7


@pytest.mark.parametrize("N,K,M", [(4096, 1024, 33920), (1024, 4096, 16960), (3072, 1024, 33920), (1024, 1024, 33920),
                                   (1472, 1472, 33920), (4416, 1472, 33920), (1472, 5888, 16960)])   # XL: ragged slabs / halves
def test_weight_gradient_with_bias_gradient_matches_reference_and_ring(N, K, M):
    """dW (N x K) = dY^T X and db = column sums of dY over M rows in split-K batches (both operands K-major, ragged K
    tiles per batch, the row sums riding on the A fragments): ops.wgrad as the training step calls it."""
    from melspec_gpt_vqvae_amd import ops

    torch.manual_seed(N + K)
    dy = (torch.randn(M, N) * 0.5).to(torch.bfloat16)
    x = (torch.randn(M, K) * 0.5).to(torch.bfloat16)
    dyd, xd = dy.to(DEV), x.to(DEV)

    def run():
        w = torch.empty(N, K, device=DEV)
        bgrad = torch.empty(N, device=DEV)
        ops.wgrad(dyd, xd, w, False, bias_out=bgrad)
        return [w, bgrad]
    ring, pp = _both(run)
    assert torch.equal(ring[0], pp[0]) and torch.equal(ring[1], pp[1])
    assert rel_err(pp[0].cpu().numpy(), (dy.float().t() @ x.float()).numpy()) < 1e-5
    assert rel_err(pp[1].cpu().numpy(), dy.float().sum(0).numpy()) < 1e-5


def test_ping_pong_loop_is_bit_reproducible():
    """LDS-DMA data is ordered for its readers only by the counted waits and barriers: a read placed too early passes
    almost every run.  (It happened: with the prologue's requests out of the stream's order one launch in a few thousand
    differed - caught by a 4 000-launch screen, tools/lab/p8_ab.py SCREEN=4000, not by twenty launches.)  2 000 launches of
    three shapes (row-major B at both tile heights, K-major B) must give the same bits every time."""
    from melspec_gpt_vqvae_amd import ops

    for M, N, K, kmaj in ((8192, 4096, 256, False), (20480, 1024, 512, False), (8192, 4096, 256, True)):
        a, b, _ = _ops(9 + N, M, N, K, b_kmajor=kmaj)
        _, p0 = _launches()
        first = ops.gemm(a, b, b_kmajor=kmaj).clone()
        same = torch.ones((), dtype=torch.bool, device=DEV)
        for _ in range(2000):
            same &= (ops.gemm(a, b, b_kmajor=kmaj) == first).all()
        assert bool(same), (M, N, K, kmaj)
        assert _launches()[1] >= p0 + 2001


def test_seeded_shape_sweep_matches_ring_bit_for_bit():
    """Ragged edges in every direction at once: M, N, K drawn from a seeded generator (N a multiple of 8 - rows are
    16-byte aligned -, K a multiple of 8, enough tiles for a full grid or, every fourth case, fewer tiles than CUs: the
    single-round form), forward / input-gradient / weight-gradient forms in turn, with and without CUs reserved for a
    collective (a grid of 240 or 248 workgroups walks different XCD blocks).  The ring kernel is the reference here; it is
    itself checked against the f32 matmul in tests/test_gemm_gpu.py."""
    import random

    from melspec_gpt_vqvae_amd import _ffi, ops

    rng = random.Random(20240)
    L = _ffi.lib()
    compared = 0
    try:
        for case in range(18):
            form = ("nt", "nn", "tn")[case % 3]
            small = case % 4 == 3
            L.melgpt_set_reserved_cus((0, 16, 8)[case % 3] if not small else 0)
            if form == "tn":       # dW (N x K) = dY^T X over M rows
                N, K, M = rng.choice([1024, 1472, 2048, 3072]), rng.choice([1024, 1472, 4096]), rng.randrange(60, 140) * 256 + rng.randrange(0, 256) // 8 * 8
                torch.manual_seed(case)
                dy = (torch.randn(M, N, device=DEV) * 0.5).to(torch.bfloat16)
                x = (torch.randn(M, K, device=DEV) * 0.5).to(torch.bfloat16)

                def run():
                    wg, bg = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
                    ops.wgrad(dy, x, wg, False, bias_out=bg)
                    return [wg, bg]
                ring, pp = _both(run)
                assert all(torch.equal(a, b) for a, b in zip(ring, pp)), (case, form, M, N, K)
                compared += 1
                continue
            N = rng.randrange(3, 18) * 256 + rng.randrange(0, 32) * 8
            K = rng.randrange(36, 200) * 8
            tiles_n = (N + 255) // 256
            rows = (90 if small else rng.randrange(300, 700)) // tiles_n + 1
            M = rows * 256 - rng.randrange(0, 255)
            torch.manual_seed(case)
            a = (torch.randn(M, K, device=DEV) * 0.5).to(torch.bfloat16)
            b = (torch.randn(*((K, N) if form == "nn" else (N, K)), device=DEV) * 0.25).to(torch.bfloat16)
            bias = torch.randn(N, device=DEV) * 0.1 if case % 2 else None
            res = torch.randn(M, N, device=DEV).to(torch.bfloat16) if case % 5 == 0 else None
            try:
                ring, pp = _both(lambda: ops.gemm(a, b, b_kmajor=form == "nn", bias=bias, residual=res))
            except AssertionError as e:
                if "meant to run" in str(e) or "takes the ring" in str(e):
                    continue        # (this draw fell to the 128 x 128 kernel: nothing to compare)
                raise
            assert torch.equal(ring, pp), (case, form, M, N, K)
            compared += 1
    finally:
        L.melgpt_set_reserved_cus(0)
    assert compared >= 12, compared
