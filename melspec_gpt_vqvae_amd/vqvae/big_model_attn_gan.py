"""VQ-VAE (SpecVQGAN-style) on the MI355X HIP kernels - host-side mirror of the reference's
vqvae/big_model_attn_gan.py: same class names, constructor/forward signatures and `state_dict` keys
(VectorQuantizer :8-71, ResnetBlock :75-135, Normalize :139-140, Downsample :145-162, nonlinearity :164-166,
Upsample :171-186, Encoder :190-282, Decoder :291-392, AttnBlock :397-450, NLayerDiscriminator :465-514,
LitVQVAE :538-634).  torch.nn.Conv2d / GroupNorm children are PARAMETER CONTAINERS only (checkpoint ABI: OIHW
f32 weights); compute runs on the C ABI with NHWC activations:
  3x3 / 1x1 convs -> implicit-GEMM MFMA kernel (melgpt_conv2d_nhwc; stride-2 pad(0,1,0,1) and nearest-x2
  upsample folded into its addressing), GroupNorm+swish -> melgpt_groupnorm_*, spatial attention -> packed
  q|k|v 1x1 GEMM + batched MFMA GEMMs + row softmax, codebook -> melgpt_vq_*.
Tensors that cross the module API are logical NCHW like the reference, but carry channels-last strides, so
chaining modules never copies.  Inference path only (the reference does not train the VQ-VAE, README.md:16);
the GAN discriminator exists so that real checkpoints load, it is not on the path.
"""
from __future__ import annotations


import torch
import torch.nn as nn

from .. import _ffi, ops
from ..flat import tensor_version
from .quantizer import VectorQuantizer  # noqa: F401  (re-exported under the reference's module path)


from . import autograd as _ag  # noqa: E402  (the differentiable path: one autograd Function per layer kind)


def _require_cuda(x):
    if not x.is_cuda:
        raise _ffi.MelgptError("melspec_gpt_vqvae_amd runs on the GPU only (no CPU / eager fallback)")


def _cdtype(module):
    return getattr(module, "compute_dtype", torch.float32)


def set_compute_dtype(module, dtype):
    assert dtype in (torch.float32, _ffi.HALF_DTYPE), f"compute dtype: float32 or {_ffi.HALF_DTYPE} (MELGPT_HALF)"
    for m in module.modules():
        object.__setattr__(m, "compute_dtype", dtype)
    return module


def _packed_weight(conv, dtype):
    """(Cout,KH,KW,Cin) copy of a Conv2d weight in the compute dtype, rebuilt only when the parameter changes."""
    w = conv.weight
    key = (tensor_version(w), dtype)
    cached = getattr(conv, "_melgpt_pack", None)
    if cached is None or cached[0] != key:
        cached = (key, ops.repack_conv_weight(w, dtype))
        object.__setattr__(conv, "_melgpt_pack", cached)
    return cached[1]


def _f32(p):
    return None if p is None else p.detach()


def _conv(conv, h, *, residual=None, stride=1, pad=None, upsample=False, out_hw=None):
    """h (B,H,W,Cin) NHWC -> conv (implicit GEMM)."""
    k = conv.kernel_size[0]
    wp = _packed_weight(conv, h.dtype)
    if pad is None:
        pad = (k // 2, k // 2)
    return ops.conv2d_nhwc(h, wp, _f32(conv.bias), stride=stride, pad=pad, upsample=upsample, residual=residual,
                           out_hw=out_hw)


class _ToNhwcFn(torch.autograd.Function):
    """logical (B,C,H,W) tensor of any strides / float dtype -> contiguous (B,H,W,C) in the compute dtype, differentiable
    (the boundary conversion of the differentiable path; the gradient comes back in the input's dtype and logical shape)."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.in_dtype = x.dtype
        return ops.to_nhwc(x, dtype)

    @staticmethod
    def backward(ctx, dy):
        g = dy.permute(0, 3, 1, 2)
        return (g if g.dtype == ctx.in_dtype else g.to(ctx.in_dtype)), None


def _to_nhwc_grad(x, dtype):
    v = x.permute(0, 2, 3, 1)
    if v.is_contiguous() and x.dtype == dtype:
        return v                      # a view: autograd follows it
    return _ToNhwcFn.apply(x, dtype)


def _as_nchw(h):
    """NHWC result -> logical NCHW view (channels-last strides)."""
    return h.permute(0, 3, 1, 2)


def nonlinearity(x):
    """swish (reference :164-166).  Stand-alone helper on the HIP path (identity-affine GroupNorm is not needed:
    implemented as a plain elementwise launch through the GELU-free cast+apply path)."""
    _require_cuda(x)
    # x * sigmoid(x) == GroupNorm-apply with mean 0, rstd 1, gamma 1, beta 0 and swish on
    h = ops.to_nhwc(x, x.dtype if x.dtype in (torch.float32, _ffi.HALF_DTYPE) else torch.float32)
    B, H, W, C = h.shape
    pad_c = (-C) % 32
    if pad_c:
        raise _ffi.MelgptError("nonlinearity(): channel count must be a multiple of 32 on the HIP path")
    dev = h.device
    zeros = torch.zeros(B * 32, device=dev)
    ones = torch.ones(B * 32, device=dev)
    y = torch.empty_like(h)
    _ffi.call("melgpt_groupnorm_apply", _ffi.ptr(h), _ffi.ptr(zeros), _ffi.ptr(ones), _ffi.ptr(torch.ones(C, device=dev)),
              _ffi.ptr(torch.zeros(C, device=dev)), _ffi.ptr(y), B, H * W, C, 1, _ffi.dtype_code(h.dtype), _ffi.stream())
    return _as_nchw(y)


def Normalize(in_channels):
    return torch.nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


def _gn(norm, h, swish):
    return ops.groupnorm(h, _f32(norm.weight), _f32(norm.bias), norm.eps, swish=swish)


def _gn_swish_conv3x3(norm, conv, h, residual=None, stats=None, next_norm=None):
    """norm -> swish -> 3x3 conv (reference :117-119 / :124-127).  One fused halo-tiled launch when the input patch
    fits LDS (the normalised tensor then never exists in HBM), else GroupNorm-apply followed by the implicit-GEMM conv.
    stats: (mean, rstd) of h for `norm` if a previous launch already produced them.  next_norm: the GroupNorm that will
    be applied to THIS conv's output - returns (y, its statistics or None) instead of y."""
    out_stats = None
    # (16-bit lane, Cout >= 256 - the 128 -> 256 block of a level change: the narrow fused kernel runs it at 530 TFLOP/s;
    # one GroupNorm-apply pass + the persistent implicit GEMM, 1 200 TFLOP/s on its ping-pong loop, take half the time)
    wide_out = h.dtype == _ffi.HALF_DTYPE and conv.out_channels >= 256
    if (conv.kernel_size[0] == 3 and conv.stride[0] == 1 and not wide_out
            and ops.fused_conv_supported(h.shape[-1], h.dtype, occupancy=2)):
        if stats is None:
            stats = ops.groupnorm_stats(h, norm.eps)
        args = (h, stats, _f32(norm.weight), _f32(norm.bias), _packed_weight(conv, h.dtype), _f32(conv.bias))
        y = None
        if next_norm is not None:
            r = ops.conv3x3_gn_with_out_stats(*args, next_norm.eps, swish=True, residual=residual)
            if r is not None:
                y, out_stats = r
        if y is None:
            y = ops.conv3x3_gn(*args, swish=True, residual=residual)
    else:
        y = _conv(conv, _gn(norm, h, True), residual=residual)
    return (y, out_stats) if next_norm is not None else y


def _block_chain(lvl, n_blocks, h, stats=None):
    """the ResnetBlocks (+ AttnBlocks) of one resolution level (reference :263-267 / :377-381).  Between two ResnetBlocks
    with nothing in between, the first one's conv2 launch also takes the GroupNorm statistics the second one's norm1
    needs (its epilogue holds the finished tile, residual included).  stats: GroupNorm statistics of h for the first
    block's norm1 when the launch that produced h took them."""
    for i_block in range(n_blocks):
        chained = len(lvl.attn) == 0 and i_block + 1 < n_blocks
        if chained:
            h, stats = lvl.block[i_block]._nhwc(h, stats=stats, next_norm=lvl.block[i_block + 1].norm1)
        else:
            h = lvl.block[i_block]._nhwc(h, stats=stats)
            stats = None
        if len(lvl.attn) > 0:
            h = lvl.attn[i_block]._nhwc(h)
    return h


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.use_conv_shortcut = conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = torch.nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if temb_channels > 0:
            self.temb_proj = torch.nn.Linear(temb_channels, out_channels)
        self.norm2 = Normalize(out_channels)
        self.dropout = torch.nn.Dropout(dropout)
        self.conv2 = torch.nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if self.in_channels != self.out_channels:
            if self.use_conv_shortcut:
                self.conv_shortcut = torch.nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
            else:
                self.nin_shortcut = torch.nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def _nhwc(self, h, stats=None, next_norm=None):
        """reference :114-135 on an NHWC tensor (temb is None on this path, dropout p = 0).  stats: GroupNorm statistics
        of h if the launch that produced h already took them; next_norm: the GroupNorm the caller applies to this
        block's output next (the following ResnetBlock's norm1) - returns (out, its statistics or None) then."""
        t, t_stats = _gn_swish_conv3x3(self.norm1, self.conv1, h, stats=stats, next_norm=self.norm2)
        if self.in_channels != self.out_channels:
            sc = _conv(self.conv_shortcut if self.use_conv_shortcut else self.nin_shortcut, h)
        else:
            sc = h
        return _gn_swish_conv3x3(self.norm2, self.conv2, t, residual=sc, stats=t_stats, next_norm=next_norm)

    def forward(self, x, temb):
        assert temb is None, "the mel VQ-VAE has no timestep embedding (temb_ch = 0, reference :196)"
        _require_cuda(x)
        if _ag.wants_grad(self, x):
            return _as_nchw(_ag.resnet_block(self, _to_nhwc_grad(x, _cdtype(self))))
        return _as_nchw(self._nhwc(ops.to_nhwc(x, _cdtype(self))))


class Downsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if self.with_conv:
            self.conv = torch.nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)
            self.pad = (0, 1, 0, 1)
        else:
            raise NotImplementedError("avg-pool downsampling is not used by the reference configuration (:229)")

    def _nhwc(self, h):
        # F.pad(x, (0,1,0,1)) + 3x3 stride-2 conv (reference :156-159): the zero row/column is an address predicate
        B, H, W, C = h.shape
        return _conv(self.conv, h, stride=2, pad=(0, 0), out_hw=((H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1))

    def forward(self, x):
        _require_cuda(x)
        if _ag.wants_grad(self, x):
            return _as_nchw(_ag.Conv3x3Fn.apply(_to_nhwc_grad(x, _cdtype(self)), self.conv.weight, self.conv.bias, None, "s2"))
        return _as_nchw(self._nhwc(ops.to_nhwc(x, _cdtype(self))))


class Upsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if not with_conv:
            raise NotImplementedError("the reference configuration always upsamples with a conv (:349)")
        self.conv = torch.nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def _nhwc(self, h):
        # nearest x2 (reference :183) folded into the conv's input addressing - the 4x larger tensor never exists
        return _conv(self.conv, h, upsample=True)

    def forward(self, x):
        _require_cuda(x)
        if _ag.wants_grad(self, x):
            return _as_nchw(_ag.Conv3x3Fn.apply(_to_nhwc_grad(x, _cdtype(self)), self.conv.weight, self.conv.bias, None, "up"))
        return _as_nchw(self._nhwc(ops.to_nhwc(x, _cdtype(self))))


class AttnBlock(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = torch.nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.k = torch.nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.v = torch.nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.proj_out = torch.nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)

    def _qkv_weights(self, dtype):
        ws = (self.q.weight, self.k.weight, self.v.weight, self.q.bias, self.k.bias, self.v.bias)
        key = tuple(tensor_version(w) for w in ws) + (dtype,)
        cached = getattr(self, "_melgpt_qkv", None)
        if cached is None or cached[0] != key:
            C = self.in_channels
            w = torch.empty(3 * C, C, dtype=dtype, device=ws[0].device)
            b = torch.empty(3 * C, dtype=torch.float32, device=ws[0].device)
            for i in range(3):
                ops.cast(ws[i].detach().reshape(C, C), dtype, out=w[i * C:(i + 1) * C])
                ops.cast(ws[3 + i].detach(), torch.float32, out=b[i * C:(i + 1) * C])
            cached = (key, w, b)
            object.__setattr__(self, "_melgpt_qkv", cached)
        return cached[1], cached[2]

    def _nhwc(self, x):
        """reference :425-450: single-head attention over the H*W positions, scale C^-1/2."""
        B, H, W, C = x.shape
        n = H * W
        dt = x.dtype
        hn = _gn(self.norm, x, False)
        wqkv, bqkv = self._qkv_weights(dt)
        kp = (n + 7) // 8 * 8      # padded key count: zero probabilities feed the K-padded P@V product
        npad = (n + 3) // 4 * 4    # score columns are written in groups of 4
        # q|k|v for all positions, plus 8 zeroed guard rows so that the per-batch (kp, C) windows below stay
        # inside this allocation and only ever meet finite values
        buf = torch.empty(B * n + 8, 3 * C, dtype=dt, device=x.device)
        ops.zero_(buf[B * n:])
        ops.gemm(hn.view(B * n, C), wqkv, bias=bqkv, out=buf[:B * n])
        win = lambda col, rows: torch.as_strided(buf, (B, rows, C), (n * 3 * C, 3 * C, 1), col * C)
        q, k, v = win(0, n), win(1, npad), win(2, kp)
        scores = torch.empty(B, n, kp, dtype=torch.float32, device=x.device)
        ops.gemm(q, k, out=scores[:, :, :npad], alpha=float(int(C) ** (-0.5)))
        probs = ops.softmax_rows(scores, n, 1.0, dt, kp)                        # (B, n, kp); columns >= n are 0
        o = ops.gemm(probs, v, b_kmajor=True)                                   # (B, n, C)
        wp = _packed_weight(self.proj_out, dt).view(C, C)
        y = ops.gemm(o.view(B * n, C), wp, bias=_f32(self.proj_out.bias), residual=x.view(B * n, C))
        return y.view(B, H, W, C)

    def forward(self, x):
        _require_cuda(x)
        if _ag.wants_grad(self, x):
            return _as_nchw(_ag.attn_block(self, _to_nhwc_grad(x, _cdtype(self))))
        return _as_nchw(self._nhwc(ops.to_nhwc(x, _cdtype(self))))


class Encoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, **ignore_kwargs):
        super().__init__()
        self.ch = ch
        self.temb_ch = 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.in_channels = in_channels
        self.conv_in = torch.nn.Conv2d(in_channels, self.ch, kernel_size=3, stride=1, padding=1)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        for i_level in range(self.num_resolutions):
            block = nn.ModuleList()
            attn = nn.ModuleList()
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for i_block in range(self.num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=self.temb_ch,
                                         dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(AttnBlock(block_in))
            down = nn.Module()
            down.block = block
            down.attn = attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res = curr_res // 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch,
                                       dropout=dropout)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch,
                                       dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = torch.nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1,
                                        padding=1)

    def _nhwc(self, x):
        """reference :254-282; x is the logical (B, in_channels, H, W) input."""
        dt = _cdtype(self)
        if self.in_channels == 1:
            B, _, H, W = x.shape
            xin = x.reshape(B, H, W)
            if not xin.is_contiguous():
                xin = xin.contiguous()
            if xin.dtype not in (torch.float32, _ffi.HALF_DTYPE):
                xin = xin.float()
            # the stem launch also takes the GroupNorm statistics the first ResnetBlock's norm1 needs (its output is the
            # largest tensor of the model: 2.2 GB at 128 tiles - no second pass over it just for 64 numbers per image)
            first_norm = self.down[0].block[0].norm1
            h, stats0 = ops.conv_in_c1(xin, self.conv_in.weight.detach(), _f32(self.conv_in.bias), dt,
                                       stats_eps=first_norm.eps if self.conv_in.out_channels == 128 else None)
        else:
            h, stats0 = _conv(self.conv_in, ops.to_nhwc(x, dt)), None
        for i_level in range(self.num_resolutions):
            lvl = self.down[i_level]
            h = _block_chain(lvl, self.num_res_blocks, h, stats=stats0 if i_level == 0 else None)
            if i_level != self.num_resolutions - 1:
                h = lvl.downsample._nhwc(h)
        h = self.mid.block_1._nhwc(h)
        h = self.mid.attn_1._nhwc(h)
        h = self.mid.block_2._nhwc(h)
        return _gn_swish_conv3x3(self.norm_out, self.conv_out, h)

    def forward(self, x):
        _require_cuda(x)
        if _ag.wants_grad(self, x):
            return _as_nchw(_ag.encoder(self, x, _cdtype(self)))
        return _as_nchw(self._nhwc(x))


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, **ignorekwargs):
        super().__init__()
        self.ch = ch
        self.temb_ch = 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.in_channels = in_channels
        self.give_pre_end = give_pre_end
        self.out_ch = out_ch
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.conv_in = torch.nn.Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch,
                                       dropout=dropout)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch,
                                       dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block = nn.ModuleList()
            attn = nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for i_block in range(self.num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=self.temb_ch,
                                         dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(AttnBlock(block_in))
            up = nn.Module()
            up.block = block
            up.attn = attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res = curr_res * 2
            self.up.insert(0, up)  # prepend to get consistent order (reference :351)
        self.norm_out = Normalize(block_in)
        self.conv_out = torch.nn.Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)

    def _conv_out_weight(self):
        w = self.conv_out.weight
        key = tensor_version(w)
        cached = getattr(self, "_melgpt_cout", None)
        if cached is None or cached[0] != key:
            cached = (key, ops.repack_conv_weight(w, torch.float32).reshape(9, w.shape[1]))
            object.__setattr__(self, "_melgpt_cout", cached)
        return cached[1]

    def _nhwc(self, z):
        """reference :361-392; z logical (B, z_channels, h, w)."""
        dt = _cdtype(self)
        self.last_z_shape = z.shape
        h = _conv(self.conv_in, ops.to_nhwc(z, dt))
        h = self.mid.block_1._nhwc(h)
        h = self.mid.attn_1._nhwc(h)
        h = self.mid.block_2._nhwc(h)
        for i_level in reversed(range(self.num_resolutions)):
            lvl = self.up[i_level]
            h = _block_chain(lvl, self.num_res_blocks + 1, h)
            if i_level != 0:
                h = lvl.upsample._nhwc(h)
        if self.give_pre_end:
            return h
        h = _gn(self.norm_out, h, True)
        if self.out_ch == 1:
            y = ops.conv_out_c1(h, self._conv_out_weight(), _f32(self.conv_out.bias), torch.float32)
            return y.unsqueeze(-1)
        return _conv(self.conv_out, h)

    def forward(self, z):
        _require_cuda(z)
        if _ag.wants_grad(self, z):
            self.last_z_shape = z.shape
            return _as_nchw(_ag.decoder(self, _to_nhwc_grad(z, _cdtype(self))))
        return _as_nchw(self._nhwc(z))


# ------------------------------------------------------------------------------ discriminator (checkpoint ABI only)
def weights_init(m):
    classname = m.__class__.__name__
    if classname.find('Conv') != -1:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif classname.find('BatchNorm') != -1:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0)


class NLayerDiscriminator(nn.Module):
    """PatchGAN discriminator (reference :465-514).  Kept ONLY so that real LitVQVAE checkpoints (22
    `discriminator.main.*` keys) load strictly; GAN training is outside the hot path."""

    def __init__(self, input_nc=3, ndf=64, n_layers=3, use_actnorm=False):
        super().__init__()
        if use_actnorm:
            raise NotImplementedError("ActNorm is not defined in the reference either (:481)")
        norm_layer = nn.BatchNorm2d
        use_bias = False
        kw, padw = 4, 1
        sequence = [nn.Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw), nn.LeakyReLU(0.2, True)]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_mult_prev = nf_mult
            nf_mult = min(2 ** n, 8)
            sequence += [nn.Conv2d(ndf * nf_mult_prev, ndf * nf_mult, kernel_size=kw, stride=2, padding=padw, bias=use_bias),
                         norm_layer(ndf * nf_mult), nn.LeakyReLU(0.2, True)]
        nf_mult_prev = nf_mult
        nf_mult = min(2 ** n_layers, 8)
        sequence += [nn.Conv2d(ndf * nf_mult_prev, ndf * nf_mult, kernel_size=kw, stride=1, padding=padw, bias=use_bias),
                     norm_layer(ndf * nf_mult), nn.LeakyReLU(0.2, True)]
        sequence += [nn.Conv2d(ndf * nf_mult, 1, kernel_size=kw, stride=1, padding=padw)]
        self.main = nn.Sequential(*sequence)

    def forward(self, input):
        raise _ffi.MelgptError("the PatchGAN discriminator is a checkpoint-compatibility container only "
                               "(VQ-VAE GAN training is outside the mel->VQ->GPT hot path)")


# reference module-level hyper-parameters (:521-530)
double_z = False
z_channels = 256
resolution = 848
in_channels = 1
out_ch = 1
ch = 128
ch_mult = [1, 1, 2, 2, 4]
num_res_blocks = 2
attn_resolutions = [53]
dropout = 0.0

try:
    import pytorch_lightning as pl

    _LitBase = pl.LightningModule
except Exception:  # pragma: no cover
    _LitBase = nn.Module


class LitVQVAE(_LitBase):
    def __init__(self, num_embeddings, embedding_dim, commitment_cost=0.25, disc_start=2001, codebook_weight=1.0,
                 disc_num_layers=3, disc_in_channels=1, disc_factor=1.0, disc_weight=1.0, use_actnorm=False,
                 disc_conditional=False, disc_ndf=64, min_adapt_weight=0.0, max_adapt_weight=1e4, learning_rate=1e-3):
        super().__init__()
        self.num_embeddings = num_embeddings
        hp = dict(ch=ch, out_ch=out_ch, ch_mult=ch_mult, num_res_blocks=num_res_blocks,
                  attn_resolutions=attn_resolutions, dropout=0.0, resamp_with_conv=True, in_channels=in_channels,
                  resolution=resolution, z_channels=z_channels, double_z=double_z)
        self._encoder = Encoder(**hp)
        self._vq_vae = VectorQuantizer(num_embeddings, embedding_dim, commitment_cost)
        self._decoder = Decoder(**hp)
        self.quant_conv = torch.nn.Conv2d(z_channels, embedding_dim, 1)
        self.post_quant_conv = torch.nn.Conv2d(embedding_dim, z_channels, 1)
        self.counts = [0 for _ in range(self.num_embeddings)]
        self.learning_rate = learning_rate
        self.codebook_weight = codebook_weight
        self.discriminator = NLayerDiscriminator(input_nc=disc_in_channels, n_layers=disc_num_layers,
                                                 use_actnorm=use_actnorm, ndf=disc_ndf).apply(weights_init)
        self.discriminator_iter_start = disc_start * 2
        self.disc_factor = disc_factor
        self.discriminator_weight = disc_weight
        self.disc_conditional = disc_conditional
        self.min_adapt_weight = min_adapt_weight
        self.max_adapt_weight = max_adapt_weight

    # The kernels address an activation through 32-bit buffer offsets: a full-resolution 128-channel 16-bit tensor may
    # hold 4 GiB, i.e. 247 tiles of 80 x 848.  Inference entry points split bigger batches into chunks of this many
    # full-resolution pixels (128 tiles of 80 x 848) - the training path is never that large (BASELINE batch 128 per GPU).
    _CHUNK_PIXELS = 128 * 80 * 848

    @classmethod
    def _chunks(cls, B, pixels_per_item):
        n = max(1, cls._CHUNK_PIXELS // max(1, pixels_per_item))
        return [(i, min(B, i + n)) for i in range(0, B, n)] if B > n else None

    def encode(self, x):
        """reference :604-608 -> z logical (B, D, 5, 53) (channels-last strides: the flat (N, D) matrix the
        codebook kernel wants)."""
        _require_cuda(x)
        if _ag.wants_grad(self, x):
            return _as_nchw(_ag.conv1x1(self.quant_conv, _ag.encoder(self._encoder, x, _cdtype(self._encoder))))
        h = self._encoder._nhwc(x)
        return _as_nchw(_conv(self.quant_conv, h))

    def decode(self, quant):
        """reference :610-614."""
        _require_cuda(quant)
        ck = None if torch.is_grad_enabled() else self._chunks(quant.shape[0], 256 * quant.shape[2] * quant.shape[3])
        if ck is not None:
            return torch.cat([self.decode(quant[a:b]) for a, b in ck], 0)
        dt = _cdtype(self._decoder)
        if _ag.wants_grad(self, quant):
            self._decoder.last_z_shape = quant.shape
            return _as_nchw(_ag.decoder(self._decoder, _ag.conv1x1(self.post_quant_conv, _to_nhwc_grad(quant, dt))))
        q = _conv(self.post_quant_conv, ops.to_nhwc(quant, dt))
        return _as_nchw(self._decoder._nhwc(_as_nchw(q)))

    @torch.no_grad()
    def encode_to_codes(self, x, fused=None):
        """mel tiles (B,1,80,848) in [-1,1] -> (B,5,53) int64 codes: the whole of extract_codes.get_codes'
        device work (feature_extraction/extract_codes.py:48-50) without the quantised tensor, loss or one-hot.
        bf16 lane (fused=None/True): the lookup runs on the encoder's output with quant_conv folded into the prepared
        codebook image - z is never formed; f32 parity lane (or fused=False): quant_conv, then the reference's
        (|x|^2 + |e|^2) - 2 x.e on the real z."""
        _require_cuda(x)
        ck = self._chunks(x.shape[0], x.shape[2] * x.shape[3])
        if ck is not None:
            return torch.cat([self.encode_to_codes(x[a:b], fused) for a, b in ck], 0)
        h = self._encoder._nhwc(x)
        if fused is None:
            fused = h.dtype == _ffi.HALF_DTYPE
        if fused:
            return self._vq_vae.encode_indices_fused(_as_nchw(h), self.quant_conv)
        return self._vq_vae.encode_indices(_as_nchw(_conv(self.quant_conv, h)))

    def forward(self, x):
        z = self.encode(x)
        loss, quantized, info = self._vq_vae(z)
        x_recon = self.decode(quantized)
        if not self.training:
            idx = info[2].squeeze().tolist()
            self.counts = [idx.count(i) + self.counts[i] for i in range(self.num_embeddings)]
        return loss, x_recon, info

    def get_input(self, batch):
        x = batch['image']
        if len(x.shape) == 3:
            x = x[..., None]
        x = x.permute(0, 3, 1, 2).to(memory_format=torch.contiguous_format)
        return x.float()
