#!/usr/bin/env python3
"""Where the Winograd kernel's output differs from the direct kernel: error by cout / row / column / image."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motif_amd import ops
from motif_amd.models.modules.layers import Conv2d
torch.manual_seed(0)
n, cin, cout, H, W = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (1, 64, 64, 16, 32))]
m = Conv2d(cin, cout, 3, 1, 1).cuda()
x = torch.randn(n, cin, H, W, device="cuda")
ops.set_conv_mma(ops.MMA_BF16X3)
ops.set_option("conv_engine", 1); ref = m(x).clone()
ops.set_option("conv_engine", 5); out = m(x).clone()
torch.cuda.synchronize()
d = (out - ref).abs()
print("max", float(d.max()), "bad fraction", float((d > 1e-4).float().mean()))
bad = d > 1e-4
print("by image ", bad.float().mean((1, 2, 3)).cpu().numpy().round(3))
print("by cout  ", bad.float().mean((0, 2, 3)).cpu().numpy().round(2))
print("by row   ", bad.float().mean((0, 1, 3)).cpu().numpy().round(2))
print("by column", bad.float().mean((0, 1, 2)).cpu().numpy().round(2))
