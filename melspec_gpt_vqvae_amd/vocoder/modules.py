"""MelGAN generator on the HIP kernels: mel (B, 80, T) -> waveform (B, 1, 256 T).
Mirror of the reference's vocoder/modules.py (:9-79): same classes, constructor arguments and - through
torch.nn.utils.weight_norm - the same state_dict keys (`model.<i>.weight_g / weight_v / bias`,
`model.<i>.block.<j>...`, `model.<i>.shortcut...`), so `best_netG.pt` checkpoints load by name
(callbacks/GPT_callbacks.py:66-79).

Execution (inference only): activations are channels-last (B, L, C).  A Conv1d with k taps and dilation d is k
batched MFMA GEMMs (melgpt_gemm, accumulate) of shifted row windows of the padded activation against the tap's
(Cout, Cin) weight slice; a ConvTranspose1d(stride r, kernel 2r) is r output phases x 2 taps of the same, written to
every r-th output row.  The LeakyReLU(0.2) in front of each convolution is applied by the padding copy
(melgpt_pad1d_act); the last Conv1d(ngf, 1, 7) + Tanh is melgpt_conv1d_out1.  No GPU work happens in torch."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch.nn.utils import weight_norm

from .. import _ffi, ops

LEAK = 0.2


def weights_init(m):
    classname = m.__class__.__name__
    if classname.find("Conv") != -1:
        m.weight.data.normal_(0.0, 0.02)
    elif classname.find("BatchNorm2d") != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


def WNConv1d(*args, **kwargs):
    return weight_norm(nn.Conv1d(*args, **kwargs))


def WNConvTranspose1d(*args, **kwargs):
    return weight_norm(nn.ConvTranspose1d(*args, **kwargs))


def _effective_weight(m):
    """g * v / ||v|| (norm over all dims but 0 - torch.nn.utils.weight_norm's default), recomputed from the parameters
    so that it is right after load_state_dict; f32, on the parameters' device."""
    g, v = m.weight_g.detach(), m.weight_v.detach()
    return torch._weight_norm(v, g, 0)


class _TapCache:
    """per-module cache of the packed weight in the compute dtype: Conv1d -> (Cout, k * Cin), the taps' (Cout, Cin) slices
    side by side; ConvTranspose1d (stride r, kernel 2r) -> ((r * Cout, 3 * Cin) phase-major matrix, bias repeated r times)"""

    def __init__(self):
        self.key, self.packed = None, None

    def get(self, m, dtype, transposed):
        key = (m.weight_g._version, m.weight_v._version, m.weight_g.data_ptr(), dtype)
        if key != self.key:
            w = _effective_weight(m)                           # Conv1d: (Cout, Cin, k); ConvTranspose1d: (Cin, Cout, k)
            if not transposed:
                packed = ops.cast(w.permute(0, 2, 1).reshape(w.shape[0], -1).contiguous(), dtype)
            else:
                # All r output phases as ONE three-tap convolution with r * Cout output channels: output row q r + s =
                # W[phi + r] x[q + c - 1] + W[phi] x[q + c], (c, phi) = divmod(s + p, r), c in {0, 1} - so phase s uses taps
                # c and c + 1 of the window (x[q - 1], x[q], x[q + 1]) and a zero block for the third; the (B, L, r * Cout)
                # result IS the (B, r L, Cout) output.  One launch that reads x once instead of r launches of two taps
                # (1.5 x the MACs, which these memory- / latency-bound layers do not notice).
                r, p = m.stride[0], m.padding[0]
                cin, cout = w.shape[0], w.shape[1]
                wall = torch.zeros(r, cout, 3, cin, dtype=w.dtype, device=w.device)
                for s in range(r):
                    c, phi = divmod(s + p, r)
                    wall[s, :, c, :] = w[:, :, phi + r].t()
                    wall[s, :, c + 1, :] = w[:, :, phi].t()
                packed = (ops.cast(wall.reshape(r * cout, 3 * cin).contiguous(), dtype),
                          m.bias.detach().float().repeat(r).contiguous() if m.bias is not None else None)
            self.key, self.packed = key, packed
        return self.packed


def _cache(m):
    c = getattr(m, "_melgpt_taps", None)
    if c is None:
        c = _TapCache()
        object.__setattr__(m, "_melgpt_taps", c)
    return c


def _conv1d(h, m, *, pad, reflect=True, leaky=False, out=None, accumulate=False, out_leaky=False):
    """h (B, L, Cin) channels-last -> (B, L, Cout): [LeakyReLU ->] [ReflectionPad1d(pad) ->] Conv1d `m` (stride 1) as ONE
    implicit-GEMM launch: the padding is an address reflection, the activation is applied to the operand on its way into
    LDS (no padded copy, no per-tap launches)."""
    k, d = m.kernel_size[0], m.dilation[0]
    assert m.stride[0] == 1 and pad * 2 == d * (k - 1), "'same' convolutions only"
    wcat = _cache(m).get(m, h.dtype, False)
    if k == 1 and not leaky and out is None and not accumulate:
        # a plain 1 x 1 convolution (the ResnetBlock shortcut) is a plain GEMM: wide stages get the persistent 256 x 256
        # kernel (dim 256 at 64 clips: 192 -> ~80 us per layer)
        B, L, Cin = h.shape
        return ops.gemm(h.view(B * L, Cin), wcat, bias=m.bias).view(B, L, wcat.shape[0])
    return ops.conv1d_nlc(h, wcat, m.bias, k, dilation=d, pad_l=pad, reflect=reflect and pad > 0,
                          in_slope=LEAK if leaky else 0.0, out=out, accumulate=accumulate,
                          out_slope=LEAK if out_leaky else 0.0)


def _conv_transpose1d(h, m):
    """LeakyReLU -> ConvTranspose1d(stride r, kernel 2r, padding r//2 + r%2, output_padding r%2): (B, L, Cin) -> (B, rL, Cout).
    Output row o = q r + s takes x[q + c] W[:, :, phi] + x[q + c - 1] W[:, :, phi + r] with (c, phi) = divmod(s + p, r): all
    phases together are one three-tap implicit GEMM with r * Cout output channels (zero rows outside the input)."""
    B, L, Cin = h.shape
    r, k, p = m.stride[0], m.kernel_size[0], m.padding[0]
    assert k == 2 * r and p == r // 2 + r % 2 and m.output_padding[0] == r % 2 and m.dilation[0] == 1
    wall, bias_r = _cache(m).get(m, h.dtype, True)
    Cout = wall.shape[0] // r
    Lout = (L - 1) * r - 2 * p + k + m.output_padding[0]
    assert Lout == r * L
    y = ops.conv1d_nlc(h, wall, bias_r, 3, pad_l=1, reflect=False, in_slope=LEAK)      # (B, L, r * Cout)
    return y.view(B, Lout, Cout)


class ResnetBlock(nn.Module):
    def __init__(self, dim, dilation=1):
        super().__init__()
        self.block = nn.Sequential(
            nn.LeakyReLU(0.2),
            nn.ReflectionPad1d(dilation),
            WNConv1d(dim, dim, kernel_size=3, dilation=dilation),
            nn.LeakyReLU(0.2),
            WNConv1d(dim, dim, kernel_size=1),
        )
        self.shortcut = WNConv1d(dim, dim, kernel_size=1)

    def _fragments(self, dtype):
        """the three weight matrices as MFMA operand fragments in lane order (csrc/vocoder.hip, resblock_narrow_kernel):
        lane = 16 g + i holds, of fragment (k-step, channel tile nt), output channel 16 nt + i and the 8 k-slots of group g"""
        conv3, conv1, sc = self.block[2], self.block[4], self.shortcut
        key = tuple((m.weight_g._version, m.weight_v._version, m.weight_g.data_ptr()) for m in (conv3, conv1, sc)) + (dtype,)
        c = getattr(self, "_melgpt_frags", None)
        if c is None or c[0] != key:
            w3, w1, ws = _effective_weight(conv3), _effective_weight(conv1)[:, :, 0], _effective_weight(sc)[:, :, 0]
            C = w3.shape[0]
            NT, KS = C // 16, C // 32
            dev = w3.device
            i = torch.arange(16, device=dev).view(1, 16, 1)
            g = torch.arange(4, device=dev).view(4, 1, 1)
            j = torch.arange(8, device=dev).view(1, 1, 8)
            frags = []
            for tap in range(3):
                for ks in range(KS):
                    for nt in range(NT):
                        frags.append(w3[16 * nt + i, 32 * ks + 8 * g + j, tap])               # (4, 16, 8) = [g][i][j]
            for ks in range(KS):
                for nt in range(NT):
                    frags.append(ws[16 * nt + i, 32 * ks + 8 * g + j])
            # conv1 contracts over leaky(t1) as the MFMA left it: k-slot j of group g = channel 4 g + j of the pair's first
            # 16-channel tile (j < 4) or 4 g + j - 4 of its second
            ch = torch.where(j < 4, 4 * g + j, 16 + 4 * g + j - 4)
            for p in range(KS):
                for nt in range(NT):
                    frags.append(w1[16 * nt + i, 32 * p + ch])
            wfrag = ops.cast(torch.stack(frags).reshape(-1, 8).contiguous(), dtype)               # (F * 64, 8)
            b3 = conv3.bias.detach().float().contiguous()
            b1s = (conv1.bias.detach().float() + sc.bias.detach().float()).contiguous()
            c = (key, wfrag, b3, b1s)
            object.__setattr__(self, "_melgpt_frags", c)
        return c[1], c[2], c[3]

    def _run(self, h):
        d = self.block[1].padding[0]
        B, L, C = h.shape
        if h.dtype != torch.float32 and d < L and ((C in (32, 64) and L % 16 == 0) or (C == 128 and L % 64 == 0)):
            # the whole block in one pass: weights as MFMA fragments in registers (dim 32 / 64) or filling the LDS (dim 128)
            wfrag, b3, b1s = self._fragments(h.dtype)
            return ops.resblock_narrow(h, wfrag, b3, b1s, d, LEAK)
        t1 = _conv1d(h, self.block[2], pad=d, leaky=True, out_leaky=True)                # leaky(conv3(pad(leaky(x))))
        y = _conv1d(h, self.shortcut, pad=0)
        # shortcut(x) + conv1(leaky t1): a plain GEMM with the shortcut as its residual operand (persistent kernel at dim 256)
        w1 = _cache(self.block[4]).get(self.block[4], h.dtype, False)
        return ops.gemm(t1.view(B * L, C), w1, bias=self.block[4].bias, residual=y.view(B * L, C)).view(B, L, C)

    def forward(self, x):  # (B, C, L) like the reference
        return _from_cl(self._run(_to_cl(x, _dtype_of(self))), x.dtype)


def _dtype_of(module):
    return getattr(module, "compute_dtype", torch.float32)


def _to_cl(x, dtype):
    """(B, C, L) -> channels-last (B, L, C) in `dtype`"""
    if not x.is_cuda:
        raise _ffi.MelgptError("melspec_gpt_vqvae_amd runs on the GPU only (no CPU / eager fallback)")
    B, C, L = x.shape
    return ops.to_nhwc(x.reshape(B, C, 1, L), dtype).view(B, L, C)


def _from_cl(h, dtype):
    B, L, C = h.shape
    return ops.to_nchw_contiguous(h.view(B, 1, L, C), dtype).view(B, C, L)


class Generator(nn.Module):
    def __init__(self, input_size, ngf, n_residual_layers):
        super().__init__()
        ratios = [8, 8, 2, 2]
        self.hop_length = np.prod(ratios)
        mult = int(2 ** len(ratios))
        model = [
            nn.ReflectionPad1d(3),
            WNConv1d(input_size, mult * ngf, kernel_size=7, padding=0),
        ]
        for i, r in enumerate(ratios):  # upsample to raw audio scale
            model += [
                nn.LeakyReLU(0.2),
                WNConvTranspose1d(mult * ngf, mult * ngf // 2, kernel_size=r * 2, stride=r, padding=r // 2 + r % 2,
                                  output_padding=r % 2),
            ]
            for j in range(n_residual_layers):
                model += [ResnetBlock(mult * ngf // 2, dilation=3 ** j)]
            mult //= 2
        model += [
            nn.LeakyReLU(0.2),
            nn.ReflectionPad1d(3),
            WNConv1d(ngf, 1, kernel_size=7, padding=0),
            nn.Tanh(),
        ]
        self.model = nn.Sequential(*model)
        self.apply(weights_init)

    @torch.no_grad()
    def forward(self, x):
        """x (B, input_size, T) -> (B, 1, 256 T) f32"""
        dt = _dtype_of(self)
        layers = list(self.model)
        h = _to_cl(x, dt)
        h = _conv1d(h, layers[1], pad=3)                      # ReflectionPad1d(3) + Conv1d(k=7)
        i = 2
        while i < len(layers):
            m = layers[i]
            if isinstance(m, nn.LeakyReLU) and isinstance(layers[i + 1], nn.ConvTranspose1d):
                h = _conv_transpose1d(h, layers[i + 1])
                i += 2
            elif isinstance(m, ResnetBlock):
                h = m._run(h)
                i += 1
            elif isinstance(m, nn.LeakyReLU) and isinstance(layers[i + 1], nn.ReflectionPad1d):
                last = layers[i + 2]
                assert isinstance(last, nn.Conv1d) and last.out_channels == 1 and isinstance(layers[i + 3], nn.Tanh)
                key = (last.weight_g._version, last.weight_v._version, last.weight_g.data_ptr())
                c = getattr(last, "_melgpt_w1", None)
                if c is None or c[0] != key:
                    c = (key, _effective_weight(last)[0].t().contiguous().reshape(-1).float())   # (k, Cin) tap-major, f32
                    object.__setattr__(last, "_melgpt_w1", c)
                w, k, pad = c[1], last.kernel_size[0], layers[i + 1].padding[0]
                if h.dtype != torch.float32 and h.shape[2] in (32, 64) and pad == k // 2 and k % 2 == 1 and pad < h.shape[1]:
                    y = ops.conv1d_out1_fused(h, w, last.bias, k, LEAK, tanh=True)        # no padded copy
                else:
                    hp = ops.pad1d_act(h, pad, reflect=True, slope=LEAK)
                    y = ops.conv1d_out1(hp, w, last.bias, h.shape[1], k, tanh=True)
                return y.view(y.shape[0], 1, y.shape[1])
            else:
                raise RuntimeError(f"unexpected layer {type(m).__name__} at model[{i}]")
        raise RuntimeError("generator has no output layer")
