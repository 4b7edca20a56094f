#!/usr/bin/env python3
"""LAB: where the 256 x 128-tile convolution differs from F.conv2d (per 256-row tile, per 16-column block)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from melspec_gpt_vqvae_amd import ops
B, H, W, Cin, Cout = int(sys.argv[1]), 80, 848, 128, 128
torch.manual_seed(11)
x = (torch.randn(B, H, W, Cin) * 0.5).to(torch.bfloat16)
w = (torch.randn(Cout, Cin, 3, 3) * 0.05).to(torch.bfloat16)
bias = torch.randn(Cout) * 0.1
ref = F.conv2d(F.pad(x.float().permute(0, 3, 1, 2), (0, 1, 0, 1)), w.float(), bias, stride=2).permute(0, 2, 3, 1)
wp = w.permute(0, 2, 3, 1).contiguous().cuda()
y = ops.conv2d_nhwc(x.cuda(), wp, bias.cuda(), stride=2, pad=(0, 0), out_hw=(40, 424)).float().cpu()
err = (y - ref).abs().reshape(-1, Cout)
M = err.shape[0]
bad_rows = (err.max(1).values > 0.05).nonzero().flatten()
print("rows", M, "bad rows", bad_rows.numel(), "max err", float(err.max()))
if bad_rows.numel():
    tiles = torch.unique(bad_rows // 256)
    print("bad tiles", tiles.tolist()[:80], "of", (M + 255) // 256)
    r0 = int(bad_rows[0])
    print("first bad row", r0, "row in tile", r0 % 256, "bad cols", (err[r0] > 0.05).nonzero().flatten().tolist()[:40])
    t0 = int(tiles[0])
    blk = err[t0 * 256:(t0 + 1) * 256]
    print("bad rows in first bad tile", (blk.max(1).values > 0.05).nonzero().flatten().tolist()[:64])
    print("bad cols in first bad tile", (blk.max(0).values > 0.05).nonzero().flatten().tolist())
