#!/usr/bin/env python3
"""Time the fused DCNv2 kernel at the shapes of the path (PCD / deformable ConvLSTM levels)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motif_amd import ops
from motif_amd.models.modules.DCNv2.dcn_v2 import DCN_sep

def main():
    reps = int(os.environ.get("REPS", "10"))
    for (n, h, w) in ((8, 180, 320), (8, 90, 160), (8, 45, 80), (2, 180, 320)):
        m = DCN_sep(64, 64, 3, stride=1, padding=1, dilation=1, deformable_groups=8).cuda()
        with torch.no_grad():
            m.conv_offset_mask.weight.normal_(0, 0.02); m.conv_offset_mask.bias.zero_()
        x = torch.randn(n, 64, h, w, device="cuda"); fea = torch.randn(n, 64, h, w, device="cuda")
        for _ in range(2): y = m(x, fea)
        om = ops.conv2d(m.conv_offset_mask.plan(), fea, act=ops.ACT_NONE, act2=ops.ACT_SIGMOID, act_split=144)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): y = ops.dcn_v2(m.dplan(), x, om)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / reps
        print("dcn (%d,64,%d,%d): %.1f us  %.1f TFLOP/s" % (n, h, w, us, 2.0 * n * 64 * 576 * h * w / us / 1e6))

if __name__ == "__main__":
    main()
