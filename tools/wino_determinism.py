#!/usr/bin/env python3
"""conv_wino.hip must give the same bits for a frame whatever the batch it sits in and however often it is run (a tile's
arithmetic does not depend on which workgroup picks it up, nor on what that workgroup did before)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motif_amd import ops
from motif_amd.models.modules.layers import Conv2d

def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale

ops.set_conv_mma(int(os.environ.get("MMA", str(ops.MMA_BF16X3))))
ops.set_option("conv_engine", int(os.environ.get("ENGINE", "5")))
bad = 0
for n, cin, cout, H, W, act, rm in [(3, 64, 64, 64, 96, ops.ACT_RELU, 2), (3, 64, 64, 64, 96, ops.ACT_NONE, 0), (6, 64, 64, 180, 320, ops.ACT_LRELU, 0),
                                    (3, 128, 64, 64, 96, ops.ACT_RELU, 0), (3, 64, 216, 64, 96, ops.ACT_NONE, 0), (5, 64, 64, 16, 24, ops.ACT_NONE, 1)]:
    m = Conv2d(cin, cout, 3, 1, 1)
    with torch.no_grad():
        m.weight.copy_(rnd(*m.weight.shape, seed=1, scale=1.0 / math.sqrt(cin * 9)))
        m.bias.copy_(rnd(cout, seed=2, scale=0.1))
    m = m.cuda()
    x, res = rnd(n, cin, H, W, seed=3).cuda(), rnd(n, cout, H, W, seed=4).cuda()
    kw = dict(act=act) if rm == 0 else dict(act=act, res=res, res_mode=rm)
    full = m(x, **kw).clone()
    for rep in range(10):
        again = m(x, **kw)
        if not torch.equal(again, full):
            d = (again != full)
            print("  run-to-run difference", (n, cin, cout, H, W), "rep", rep, int(d.sum()), "values; first", d.nonzero()[:4].tolist()); bad += 1
            break
    for k in range(1, n):
        kw2 = dict(act=act) if rm == 0 else dict(act=act, res=res[:k].contiguous(), res_mode=rm)
        part = m(x[:k].contiguous(), **kw2)
        if not torch.equal(part, full[:k]):
            d = (part != full[:k])
            idx = d.nonzero()
            print("  batch-dependent result", (n, cin, cout, H, W), "act", act, "res_mode", rm, "k", k, int(d.sum()), "values; max |diff| %.3e" % float((part - full[:k]).abs().max()),
                  "first", idx[:3].tolist(), "rows", sorted(set(idx[:, 2].tolist()))[:12], "cols", sorted(set(idx[:, 3].tolist()))[:12]); bad += 1
print("DETERMINISM", "OK" if not bad else "FAILED (%d)" % bad)
