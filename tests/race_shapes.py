"""The launch forms of the hand-synchronised kernels that tests/test_race_screens_gpu.py repeats and tests/race_worker.py
replays on the vmcnt(0) build (test infrastructure).  Each entry: name -> builder(torch, ops, dev) returning fn with
fn() -> list of output tensors.  Seeded inputs, so the parent (production library) and the child (race-screen build of
the same sources, csrc/common.h MELGPT_VMCNT0) see the same operands.

Kernels and why they are here: gemm8p_kernel (csrc/gemm8p.hip: LDS-DMA ordered by ONE counted vmcnt per K tile + raw
s_barriers), gemm256_kernel (csrc/gemm256.hip: ring of five half-unit slots, counted wait per K unit;
tickets through a global-memory mailbox), conv3x3_gn_ws_kernel (csrc/conv_fused.hip: staging waves publish LDS counters
behind counted waits), attn_q_kernel / attn_bwd1_kernel (csrc/attn.hip: LDS work counters, hand-placed waits).
Reference call sites: transformer/minGPT.py:76-88,100-117 (the Linear layers and attention of a Block),
vqvae/big_model_attn_gan.py:75-135 (ResnetBlock's norm -> swish -> conv)."""


def _rnd(torch, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    return lambda *s, scale=0.5, dtype=torch.bfloat16: (torch.randn(*s, device=dev, generator=g) * scale).to(dtype)


def _gemm(M, N, K, *, kmaj=False, seed=1, **epi):
    def build(torch, ops, dev):
        r = _rnd(torch, dev, seed)
        a = r(M, K)
        b = r(K, N, scale=0.25) if kmaj else r(N, K, scale=0.25)
        kw = {}
        if epi.get("bias"):
            kw["bias"] = r(N, scale=0.1, dtype=torch.float32)
        if epi.get("residual") or epi.get("mul"):
            kw["residual"] = r(M, N, scale=1.0)
        if epi.get("drop"):
            kw.update(drop_p=0.25, seed=77, stream_id=5)
        if epi.get("mul"):
            kw["act"] = ops.ACT_MUL
        if epi.get("dact"):
            def fn():
                pre = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                return [ops.gemm(a, b, act=ops.ACT_GELU_DACT, pre_out=pre, **kw), pre]
            return fn
        return lambda: [ops.gemm(a, b, b_kmajor=kmaj, **kw)]
    return build


def _wgrad(M, N, K, seed=2):
    def build(torch, ops, dev):
        r = _rnd(torch, dev, seed)
        dy, x = r(M, N), r(M, K)

        def fn():
            w, bg = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
            ops.wgrad(dy, x, w, False, bias_out=bg)
            return [w, bg]
        return fn
    return build


def _switched(build, setter, value):
    """the same form with a library switch held at `value` for every call (melgpt_set_attn_fwd32)"""
    def build2(torch, ops, dev):
        from melspec_gpt_vqvae_amd import _ffi

        fn = build(torch, ops, dev)
        set_ = getattr(_ffi.lib(), setter)

        def held():
            prev = set_(value)
            try:
                return fn()
            finally:
                set_(prev)
        return held
    return build2


def _conv(B, H, W, *, stats_out=False, residual=False, seed=3):
    def build(torch, ops, dev):
        C = 128
        r = _rnd(torch, dev, seed)
        x = r(B, H, W, C, scale=1.0)
        res = r(B, H, W, C, scale=1.0) if residual else None
        w = r(C, 3, 3, C, scale=0.05)
        bias = r(C, scale=0.1, dtype=torch.float32)
        gamma = r(C, scale=0.2, dtype=torch.float32) + 1.0
        beta = r(C, scale=0.1, dtype=torch.float32)
        st = ops.groupnorm_stats(x, 1e-6)
        if stats_out:
            def fn():
                y, (m, s) = ops.conv3x3_gn_with_out_stats(x, st, gamma, beta, w, bias, 1e-6, swish=True, residual=res)
                return [y, m, s]
            return fn
        return lambda: [ops.conv3x3_gn(x, st, gamma, beta, w, bias, swish=True, residual=res)]
    return build


def _conv_plain(B, H, W, Cin, Cout, *, stride=1, upsample=False, residual=False, seed=40, k=3):
    """implicit-GEMM convolution without a norm in front (Downsample / Upsample: big_model_attn_gan.py:145-186)"""
    def build(torch, ops, dev):
        r = _rnd(torch, dev, seed)
        x = r(B, H, W, Cin, scale=1.0)
        w = r(Cout, k, k, Cin, scale=0.05)
        bias = r(Cout, scale=0.1, dtype=torch.float32)
        oh = (H // 2, W // 2) if stride == 2 else ((2 * H, 2 * W) if upsample else (H, W))
        res = r(B, oh[0], oh[1], Cout, scale=1.0) if residual else None
        pad = (0, 0) if stride == 2 else (k // 2, k // 2)
        return lambda: [ops.conv2d_nhwc(x, w, bias, stride=stride, pad=pad, out_hw=oh, upsample=upsample, residual=res)]
    return build


def _attn(B, H, T, p_drop, n_unmasked=0, bwd=False, seed=5):
    def build(torch, ops, dev):
        C = 64 * H
        r = _rnd(torch, dev, seed)
        q, k, v, do = r(B * T, C), r(B * T, C), r(B * T, C), r(B * T, C)
        kw = dict(B=B, T=T, n_unmasked=n_unmasked, drop_p=p_drop, seed=1, stream_id=0)
        o, lse, _ = ops.attn_fwd(q, k, v, H, **kw)
        o, lse = o.clone(), lse.clone()
        if bwd:
            return lambda: list(ops.attn_bwd(q, k, v, o, do, lse, H, **kw))
        return lambda: list(ops.attn_fwd(q, k, v, H, **kw)[:2])
    return build


# name -> (kernel family, builder).  GEMM shapes: >= 256 tiles unless marked; short K loops keep a launch at 20-80 us.
FORMS = {
    # ---- gemm8p: three operand forms x both tile heights x EDGE x split-K x single round x half-height tail
    "gemm NT 8192x4096x256 (256-row tiles)": ("gemm8p", _gemm(8192, 4096, 256)),
    "gemm NT 20480x1024x512 (192-row tiles)": ("gemm8p", _gemm(20480, 1024, 512, seed=11)),
    "gemm NN 8192x4096x256 (K-major B)": ("gemm8p", _gemm(8192, 4096, 256, kmaj=True, seed=12)),
    "gemm NN 24576x1024x320 (K-major B, 192-row tiles, ragged K)": ("gemm8p", _gemm(24576, 1024, 320, kmaj=True, seed=13)),
    "gemm NN 16960x1472x328 (EDGE: N ends inside a half-tile)": ("gemm8p", _gemm(16960, 1472, 328, kmaj=True, seed=14)),
    "gemm NT 33920x4096x264 (half-height last round, ragged K)": ("gemm8p", _gemm(33920, 4096, 264, seed=15)),
    "gemm NT 4000x2048x512 (single round: fewer tiles than CUs)": ("gemm8p", _gemm(4000, 2048, 512, seed=16)),
    "gemm NT 8192x4096x256 + bias + GELU + derivative": ("gemm8p", _gemm(8192, 4096, 256, seed=17, bias=True, dact=True)),
    "gemm NT 20480x1024x512 + bias + dropout + residual": ("gemm8p", _gemm(20480, 1024, 512, seed=18, bias=True, drop=True, residual=True)),
    "gemm NN 8192x4096x256 x saved derivative": ("gemm8p", _gemm(8192, 4096, 256, kmaj=True, seed=19, mul=True)),
    "wgrad 4096x1024 over 8480 rows (split-K batches + bias row sums)": ("gemm8p", _wgrad(8480, 4096, 1024)),
    "wgrad 1472x4416 over 4240 rows (EDGE A and B)": ("gemm8p", _wgrad(4240, 1472, 4416, seed=21)),
    # ---- gemm8p, 256 x 128 tiles (convolutions with <= 128 output channels): block lists over several rounds, single round
    "conv3x3 s2 6x80x848 128->128 (256 x 128 tiles, two rounds)": ("gemm8p", _conv_plain(6, 80, 848, 128, 128, stride=2)),
    "conv3x3 s2 3x80x848 128->128 + residual (256 x 128 tiles, single round)":
        ("gemm8p", _conv_plain(3, 80, 848, 128, 128, stride=2, residual=True, seed=41)),
    "conv3x3 x2-upsampled 4x40x212 128->64 (256 x 128 tiles, half the columns)":
        ("gemm8p", _conv_plain(4, 40, 212, 128, 64, upsample=True, seed=42)),
    "conv1x1 10x40x212 128->256 (two K tiles per tile)": ("gemm8p", _conv_plain(10, 40, 212, 128, 256, k=1, seed=43)),
    # ---- fused GroupNorm + swish + conv3x3, wave-specialised: both tile shapes, with / without output statistics
    "conv3x3+gn 6x80x848 (16x16 tiles)": ("conv_ws", _conv(6, 80, 848)),
    "conv3x3+gn 6x80x848 + residual + output statistics": ("conv_ws", _conv(6, 80, 848, stats_out=True, residual=True, seed=31)),
    "conv3x3+gn 12x40x424 (8x32 tiles)": ("conv_ws", _conv(12, 40, 424, seed=32)),
    "conv3x3+gn 12x40x424 + output statistics": ("conv_ws", _conv(12, 40, 424, stats_out=True, seed=33)),
    # ---- attention (forward: the 16-row and the 32-row kernel, each forced; backward in one launch for the causal 16-bit lane)
    "attention forward 32x16x265 dropout 0.5 (16-row kernel)": ("attn", _switched(_attn(32, 16, 265, 0.5), "melgpt_set_attn_fwd32", 0)),
    "attention forward 32x16x265 dropout 0.5 (32-row kernel)": ("attn", _switched(_attn(32, 16, 265, 0.5, seed=50), "melgpt_set_attn_fwd32", 1)),
    "attention forward 32x16x265 dropout 0, bidirectional (32-row kernel)": ("attn", _attn(32, 16, 265, 0.0, n_unmasked=265, seed=51)),
    "attention backward 32x16x265 dropout 0.5 (single pass)": ("attn", _attn(32, 16, 265, 0.5, bwd=True, seed=52)),
    "attention backward 32x16x265 dropout 0 (single pass)": ("attn", _attn(32, 16, 265, 0.0, bwd=True, seed=53)),
    "attention backward 16x23x265 bidirectional, dropout 0.3": ("attn", _attn(16, 23, 265, 0.3, n_unmasked=265, bwd=True, seed=54)),
}
# the same forms on the RING K loop (melgpt_set_gemm_pingpong(0): the loop eval-time full epilogues and small-K shapes still take)
RING_FORMS = ["gemm NT 8192x4096x256 (256-row tiles)", "gemm NN 24576x1024x320 (K-major B, 192-row tiles, ragged K)",
              "wgrad 4096x1024 over 8480 rows (split-K batches + bias row sums)"]


def checksum(torch, t):
    """two 64-bit sums over the raw bits of t (position-weighted): equal tensors <=> equal pairs, for all practical purposes"""
    b = t.detach().contiguous().view(-1)
    b = b.view(torch.int16 if b.element_size() == 2 else torch.int32).to(torch.int64)
    w = torch.arange(b.numel(), device=b.device, dtype=torch.int64) % 65521 + 1
    return [int(b.sum()), int((b * w).sum())]
