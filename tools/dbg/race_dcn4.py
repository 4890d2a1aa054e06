import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from motif_amd import ops
from motif_amd.models.modules.layers import Conv2d
from motif_amd.models.modules.DCNv2.dcn_v2 import DCN_sep
torch.manual_seed(0)
ops.set_mma(__import__("os").environ.get("SIDE_MMA", "bf16x3"))
gru = Conv2d(242, 96, 3, 1, 1).cuda(); xg = torch.randn(2, 242, 90, 160, device="cuda")
c7 = Conv2d(3, 32, 7, 2, 3).cuda(); x7 = torch.randn(4, 3, 720, 1280, device="cuda")
tr = Conv2d(64, 64, 3, 1, 1).cuda(); xt = torch.randn(4, 64, 360, 640, device="cuda")
dcns = [DCN_sep(64, 64, 3, stride=1, padding=1, dilation=1, deformable_groups=8).cuda() for _ in range(2)]
with torch.no_grad():
    for d in dcns: d.conv_offset_mask.weight.normal_(0, 0.05)
res = {}
for (h, w) in ((180, 320),):
    xs = [torch.randn(1, 64, h, w, device="cuda") for _ in range(2)]
    feas = [torch.randn(1, 64, h, w, device="cuda") for _ in range(2)]
    oms = ops.conv2d_multi([d.conv_offset_mask.plan() for d in dcns], feas, act=ops.ACT_NONE, act2=ops.ACT_SIGMOID, act_split=144)
    oms = [oms[0].clone(), oms[1].clone()]
    ref = ops.dcn_v2_multi([d.dplan() for d in dcns], xs, oms, 8, ops.ACT_LRELU).clone()
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    bad = 0
    for it in range(int(os.environ.get("ITERS", "200"))):
        if not __import__("os").environ.get("SERIAL"):
            with torch.cuda.stream(sb):
                which = __import__("os").environ.get("SIDE", "gru,tr,c7").split(",")
                if "gru" in which: y1 = gru(xg, act=1)
                if "tr" in which: y2 = tr(xt, act=1)
                if "c7" in which: y3 = c7(x7, act=1)
        with torch.cuda.stream(sa):
            outs = [ops.dcn_v2_multi([d.dplan() for d in dcns], xs, oms, 8, ops.ACT_LRELU) for _ in range(4)]
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(o, ref)) for o in outs)
    print("dcn P=2 %dx%d: %d mismatching outputs under concurrency" % (h, w, bad))
