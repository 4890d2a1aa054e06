#!/usr/bin/env python3
"""Flow-head shaped layers (few couts, long reduction) in isolation: conv_direct.hip's deep form against the MFMA engine."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motif_amd import ops
from motif_amd.models.modules.layers import Conv2d

SHAPES = [(1, 529, 2, 12, 20), (1, 661, 2, 24, 40), (1, 629, 2, 48, 80), (1, 597, 2, 96, 160), (1, 565, 2, 192, 320), (2, 128, 2, 90, 160)]
for n, ci, co, h, w in SHAPES:
    m = Conv2d(ci, co, 3, 1, 1).cuda()
    x = torch.randn(n, ci, h, w, device="cuda")
    row = []
    for nodirect in (0, 1):
        ops.set_option("conv_nodirect", nodirect)
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            m(x)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) * 1000 / 20)
    ops.set_option("conv_nodirect", 0)
    print("%s  deep %.1f us   mfma engine %.1f us" % ((n, ci, co, h, w), row[0], row[1]))
