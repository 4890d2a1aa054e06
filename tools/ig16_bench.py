#!/usr/bin/env python3
"""conv_ig16.hip against the fp32 engine (option conv_engine = 6) on the non-3x3-stride-1 layer shapes of the clip and of PWC-Net:
per shape the time of one launch under mma = 7 with and without the new kernel.  -> which shapes the dispatch should give it."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motif_amd import ops
from motif_amd.models.modules.layers import Conv2d

SHAPES = [  # cin, cout, k, stride, pad, dil, groups, H, W, N, tag
    (3, 32, 7, 2, 3, 1, 1, 720, 1280, 2, "RAFT stem"),
    (2, 64, 7, 1, 3, 1, 1, 90, 160, 2, "convf1"),
    (64, 32, 3, 1, 1, 1, 1, 90, 160, 2, "convf2"),
    (64, 64, 3, 2, 1, 1, 1, 180, 320, 8, "pyramid L2 s2 x8"),
    (64, 64, 3, 2, 1, 1, 1, 90, 160, 8, "pyramid L3 s2 x8"),
    (64, 64, 3, 2, 1, 1, 1, 180, 320, 2, "pyramid L2 s2 x2"),
    (8, 8, 3, 1, 1, 1, 1, 360, 640, 2, "bneck 8->8"),
    (16, 16, 3, 2, 1, 1, 1, 360, 640, 2, "bneck 16->16 s2"),
    (16, 16, 3, 1, 1, 1, 1, 180, 320, 2, "bneck 16->16"),
    (24, 24, 3, 2, 1, 1, 1, 180, 320, 2, "bneck 24->24 s2"),
    (24, 24, 3, 1, 1, 1, 1, 90, 160, 2, "bneck 24->24"),
    (32, 64, 1, 2, 0, 1, 1, 360, 640, 2, "down 1x1 s2"),
    (64, 96, 1, 2, 0, 1, 1, 180, 320, 2, "down 1x1 s2"),
    (14, 64, 3, 1, 1, 1, 2, 180, 320, 2, "flow_process0 g2"),
    (64, 64, 3, 1, 1, 1, 2, 180, 320, 2, "flow_process1 g2"),
    (3, 64, 3, 1, 1, 1, 1, 180, 320, 2, "conv_first"),
    (96, 160, 1, 1, 0, 1, 1, 90, 160, 2, "cnet head 1x1"),
    (3, 16, 3, 2, 1, 1, 1, 768, 1280, 2, "PWC ext1 s2"),
    (16, 16, 3, 1, 1, 1, 1, 384, 640, 2, "PWC ext1"),
    (16, 32, 3, 2, 1, 1, 1, 384, 640, 2, "PWC ext2 s2"),
    (32, 32, 3, 1, 1, 1, 1, 192, 320, 2, "PWC ext2"),
    (32, 64, 3, 2, 1, 1, 1, 192, 320, 2, "PWC ext3 s2"),
    (64, 96, 3, 2, 1, 1, 1, 96, 160, 2, "PWC ext4 s2"),
    (96, 128, 3, 2, 1, 1, 1, 48, 80, 2, "PWC ext5 s2"),
    (128, 196, 3, 2, 1, 1, 1, 24, 40, 2, "PWC ext6 s2"),
    (128, 128, 3, 1, 2, 2, 1, 192, 320, 1, "PWC refiner d2"),
    (128, 128, 3, 1, 4, 4, 1, 192, 320, 1, "PWC refiner d4"),
    (128, 96, 3, 1, 8, 8, 1, 192, 320, 1, "PWC refiner d8"),
    (96, 64, 3, 1, 16, 16, 1, 192, 320, 1, "PWC refiner d16"),
    (64, 32, 3, 1, 1, 1, 1, 192, 320, 1, "PWC refiner 64->32"),
]


def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1000.0 * e0.elapsed_time(e1) / reps


def main():
    ops.set_mma("f16x2")
    print("%-22s %-34s %12s %12s %12s   %s" % ("layer", "cin,cout,k,s,p,d,g @ HxW xN", "ig16 us", "default us", "fp32 eng us", "default dispatch"))
    for cin, cout, k, st, pd, dl, g, H, W, N, tag in SHAPES:
        m = Conv2d(cin, cout, k, st, pd, dl, g).cuda()
        x = torch.randn(N, cin, H, W, device="cuda")
        res = {}
        outs = {}
        for eng in (7, 0, 6):                            # conv_ig16.hip wherever the shape fits | the library's rule | the fp32 engine
            ops.set_option("conv_engine", eng)
            outs[eng] = m(x, act=ops.ACT_RELU)
            res[eng] = t(lambda: m(x, act=ops.ACT_RELU))
        ops.set_option("conv_engine", 0)
        took = "conv_ig16" if not torch.equal(outs[0], outs[6]) else "fp32 engine / direct"
        print("%-22s %-34s %12.1f %12.1f %12.1f   %s" % (tag, "%d,%d,%d,%d,%d,%d,%d @ %dx%d x%d" % (cin, cout, k, st, pd, dl, g, H, W, N), res[7], res[0], res[6], took), flush=True)


if __name__ == "__main__":
    main()
