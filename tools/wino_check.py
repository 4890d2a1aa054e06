#!/usr/bin/env python3
"""Round-4 Winograd F(2,3) conv kernel: error against fp64 beside the other engines, then timing per shape (engine 5 vs 6)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from motif_amd import ops
from motif_amd.models.modules.layers import Conv2d

def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale

CASES = [tuple(int(v) for v in c.split("x")) for c in os.environ["CASES"].split(",")] if os.environ.get("CASES") else [(1, 64, 64, 16, 32), (2, 64, 64, 45, 80), (1, 128, 64, 23, 36), (1, 64, 216, 20, 36), (2, 81, 96, 12, 16), (3, 64, 64, 180, 320), (1, 48, 80, 19, 36)]
for n, cin, cout, H, W in CASES:
    m = Conv2d(cin, cout, 3, 1, 1)
    with torch.no_grad():
        m.weight.copy_(rnd(*m.weight.shape, seed=1, scale=1.0 / math.sqrt(cin * 9)))
        m.bias.copy_(rnd(cout, seed=2, scale=0.1))
    x, res = rnd(n, cin, H, W, seed=3), rnd(n, cout, H, W, seed=4)
    ref = F.relu(F.conv2d(x.double(), m.weight.double(), m.bias.double(), 1, 1)) + res.double()
    m = m.cuda()
    errs = {}
    for name, mma, eng in (("fp32", ops.MMA_FP32, 0), ("split", ops.MMA_BF16X3, 1), ("split2", ops.MMA_BF16X3, 2), ("wino", ops.MMA_BF16X3, 5), ("wino_f16x2", ops.MMA_F16X2, 5)):
        ops.set_conv_mma(mma)
        ops.set_option("conv_engine", eng)
        out = m(x.cuda(), act=ops.ACT_RELU, res=res.cuda(), res_mode=2)
        torch.cuda.synchronize()
        d = (out.double().cpu() - ref).abs()
        errs[name] = (float(d.max()), float(d.mean()))
    ops.set_option("conv_engine", 0)
    print((n, cin, cout, H, W), "scale %.2f" % float(ref.abs().max()), " ".join("%s max %.2e mean %.2e |" % (k, *v) for k, v in errs.items()), flush=True)
