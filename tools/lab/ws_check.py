#!/usr/bin/env python3
"""Lab check of the wave-specialised fused conv: plain / +stats / +res variants against torch (f32 reference on the same
bf16 operands), run-to-run determinism, and where the wrong pixels are."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from melspec_gpt_vqvae_amd import ops

torch.manual_seed(0)
B, H, W, C = int(os.environ.get("B", 8)), int(os.environ.get("H", 80)), int(os.environ.get("W", 848)), 128
x = (torch.randn(B, H, W, C, device="cuda") * 1.1 + 0.3).bfloat16()
res = (torch.randn(B, H, W, C, device="cuda") * 0.7).bfloat16()
w = (torch.randn(C, C, 3, 3, device="cuda") * 0.03).bfloat16()
wp = w.permute(0, 2, 3, 1).contiguous()
bias = torch.randn(C, device="cuda") * 0.2
gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
stats = ops.groupnorm_stats(x, 1e-6)
h = F.group_norm(x.float().permute(0, 3, 1, 2), 32, gamma, beta, eps=1e-6)
h = (h * torch.sigmoid(h)).bfloat16().float()
ref = F.conv2d(h, w.float(), bias, padding=1).permute(0, 2, 3, 1)


def report(name, y, r):
    d = (y.float() - r).abs()
    bad = d > 0.05 * r.abs().max()
    print(f"{name}: max err {float(d.max()):.4f} (ref max {float(r.abs().max()):.3f}), bad elements {int(bad.sum())}")
    if bad.any():
        idx = bad.nonzero()
        bs, ys, xs, cs = idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]
        print("   images", bs.unique().tolist()[:10], "rows", ys.unique().tolist()[:20], "cols(first)", xs.unique().tolist()[:20],
              "channels(first)", cs.unique().tolist()[:20], " tiles (y/16,x/16):", torch.stack([ys // 16, xs // 16], 1).unique(dim=0).tolist()[:12])


y1 = ops.conv3x3_gn(x, stats, gamma, beta, wp, bias, swish=True)
y2 = ops.conv3x3_gn(x, stats, gamma, beta, wp, bias, swish=True)
print("plain deterministic:", bool(torch.equal(y1, y2)))
report("plain", y1, ref)
r = ops.conv3x3_gn_with_out_stats(x, stats, gamma, beta, wp, bias, 1e-6, swish=True)
report("+stats", r[0], ref)
print("stats == plain:", bool(torch.equal(r[0], y1)))
y3 = ops.conv3x3_gn(x, stats, gamma, beta, wp, bias, swish=True, residual=res)
report("+res", y3, ref + res.float())

# ---- where are the wrong values: by tile ordinal of the workgroup, position in the tile, channel
d = (y1.float() - ref).abs() > 0.05 * ref.abs().max()
tx, ty = (W + 15) // 16, (H + 15) // 16
bb, yy, xx, cc = d.nonzero(as_tuple=True)
tile = (bb * ty + yy // 16) * tx + xx // 16
print("bad by tile ordinal k = tile // 256:", torch.bincount(tile // 256, minlength=8).tolist())
print("bad by tile % 8:", torch.bincount(tile % 8, minlength=8).tolist())
print("bad by row in tile:", torch.bincount(yy % 16, minlength=16).tolist())
print("bad by col in tile:", torch.bincount(xx % 16, minlength=16).tolist())
print("bad by channel // 16:", torch.bincount(cc // 16, minlength=8).tolist())
t0 = d[0, :16, :16].float().mean().item(), d[0, :16, 16:32].float().mean().item()
print("bad fraction tile 0, tile 1:", t0, " per-tile bad fraction histogram (first 24 tiles):",
      [round(d[0, (k // tx) * 16:(k // tx) * 16 + 16, (k % tx) * 16:(k % tx) * 16 + 16].float().mean().item(), 2) for k in range(24)])

# ---- what does the wrong tile hold?  (tile 256 = first tile of ordinal 1)
k = 256
b0, r0_ = k // (tx * ty), k % (tx * ty)
ys, xs = (r0_ // tx) * 16, (r0_ % tx) * 16
got = y1[b0, ys:ys + 16, xs:xs + 16].float()
def conv_part(lo, hi):
    return F.conv2d(h[b0:b0 + 1, lo:hi], w.float()[:, lo:hi], None, padding=1).permute(0, 2, 3, 1)[0, ys:ys + 16, xs:xs + 16]
full = ref[b0, ys:ys + 16, xs:xs + 16]
p0, p1 = conv_part(0, 64), conv_part(64, 128)
for name, cand in (("full", full), ("bias + half0 only", p0 + bias), ("bias + half1 only", p1 + bias), ("bias only", bias.expand_as(full))):
    print(f"tile 256 vs {name}: max |diff| {float((got - cand).abs().max()):.4f}")
# previous tile's half 0 with this tile's half 1?
kp = 0
bp, rp = kp // (tx * ty), kp % (tx * ty)
yp, xp = (rp // tx) * 16, (rp % tx) * 16
p0_prev = F.conv2d(h[bp:bp + 1, 0:64], w.float()[:, 0:64], None, padding=1).permute(0, 2, 3, 1)[0, yp:yp + 16, xp:xp + 16]
print(f"tile 256 vs bias + half0(tile 0) + half1(tile 256): {float((got - (p0_prev + p1 + bias)).abs().max()):.4f}")
print("second run of the same conv, tile 256 equal:", bool(torch.equal(y1[b0, ys:ys + 16, xs:xs + 16], y2[b0, ys:ys + 16, xs:xs + 16])))

# ---- determinism by tile ordinal; raw-copy variant (WS_LAB & 1 builds: no affine / swish) against a raw conv
dd = (y1 != y2)
bb2, yy2, xx2, cc2 = dd.nonzero(as_tuple=True)
tile2 = (bb2 * ty + yy2 // 16) * tx + xx2 // 16
print("run-to-run differing elements by tile ordinal:", torch.bincount(tile2 // 256, minlength=8).tolist())
if os.environ.get("REFRAW"):
    rraw = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias, padding=1).permute(0, 2, 3, 1)
    d3 = (y1.float() - rraw).abs() > 0.05 * rraw.abs().max()
    b3, y3_, x3, c3 = d3.nonzero(as_tuple=True)
    t3 = (b3 * ty + y3_ // 16) * tx + x3 // 16
    print("RAW reference: bad by tile ordinal:", torch.bincount(t3 // 256, minlength=8).tolist(), "max err", float((y1.float() - rraw).abs().max()))
