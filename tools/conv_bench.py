#!/usr/bin/env python3
"""Time selected conv-engine shapes in isolation (also the target of rocprofv3 --pmc runs)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motif_amd.models.modules.layers import Conv2d

SHAPES = [  # N, Cin, Cout, k, stride, H, W
    (3, 64, 64, 3, 1, 180, 320),
    (1, 128, 64, 3, 1, 180, 320),
    (1, 128, 64, 3, 1, 90, 160),
    (1, 64, 216, 3, 1, 180, 320),
    (1, 128, 64, 3, 1, 45, 80),
    (2, 64, 64, 3, 1, 90, 160),
    (8, 64, 64, 3, 1, 180, 320),
    (8, 64, 64, 3, 1, 45, 80),
    (8, 64, 216, 3, 1, 45, 80),
    (8, 64, 64, 3, 1, 90, 160),
    (2, 64, 64, 3, 1, 90, 160),
    (8, 128, 64, 3, 1, 180, 320),
    (8, 64, 216, 3, 1, 180, 320),
    (2, 242, 96, 3, 1, 90, 160),
    (2, 64, 64, 3, 1, 180, 320),
    (2, 128, 256, 3, 1, 180, 320),
    (8, 128, 64, 3, 1, 90, 160),
    (8, 64, 216, 3, 1, 90, 160),
]

PW_SHAPES = [  # the clip's 1x1 layers (conv_pw.hip under MMA=7, the fp32 engine under MMA=0): PW=1 selects this list
    (4, 128, 64, 1, 1, 180, 320), (2, 196, 96, 1, 1, 90, 160), (1, 128, 64, 1, 1, 180, 320), (2, 24, 96, 1, 1, 90, 160),
    (2, 16, 64, 1, 1, 180, 320), (2, 64, 64, 1, 1, 180, 320), (2, 64, 96, 1, 1, 90, 160), (2, 8, 32, 1, 1, 360, 640),
]


def main():
    global SHAPES
    if os.environ.get("PW"):
        SHAPES = PW_SHAPES
    reps = int(os.environ.get("REPS", "20"))
    only = os.environ.get("ONLY")
    from motif_amd import ops
    if os.environ.get("MMA"):
        ops.set_conv_mma(int(os.environ["MMA"]))                       # 6 = three bf16 parts, 7 = two fp16 parts (conv_wino.hip)
    if os.environ.get("ENGINE"):
        ops.set_option("conv_engine", int(os.environ["ENGINE"]))      # 1 = round-2 two-block kernel, 2 / 3 = round-3 kernel, 5 = round-4 Winograd kernel, 6 = never it, 0 = the library's choice
    if os.environ.get("WINO_RPRE"):
        ops.set_option("conv_wino_rpre", int(os.environ["WINO_RPRE"]))    # 1 = residual quads requested in the epilogue only (round 4 behaviour)
    if os.environ.get("WINO_TR"):
        ops.set_option("conv_wino_tr", int(os.environ["WINO_TR"]))    # 1 = transposed accumulators + register-only epilogue (experimental), 0 = row-major (default)
    print("conv mma mode", ops.get_conv_mma(), "engine", ops.get_option("conv_engine"), "wino_tr", ops.get_option("conv_wino_tr"))
    for i, (n, ci, co, k, s, h, w) in enumerate(SHAPES):
        if only is not None and int(only) != i:
            continue
        m = Conv2d(ci, co, k, s, k // 2).cuda()
        x = torch.randn(n, ci, h, w, device="cuda")
        res = torch.randn(n, co, h // s, w // s, device="cuda") if os.environ.get("RES") else None
        kw = dict(act=int(os.environ.get("ACT", "1")), res=res, res_mode=1) if res is not None else dict(act=int(os.environ.get("ACT", "1")))
        if os.environ.get("ACT_SPLIT"):                     # the offset | sigmoid(mask) layer of a DCN: ACT=0 ACT_SPLIT=144
            kw.update(act2=ops.ACT_SIGMOID, act_split=int(os.environ["ACT_SPLIT"]))
        for _ in range(3):
            y = m(x, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            y = m(x, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / reps
        fl = 2.0 * n * co * ci * k * k * (h // s) * (w // s)
        print("shape %d %s: %.1f us  %.1f TFLOP/s" % (i, (n, ci, co, k, s, h, w), us, fl / us / 1e6), flush=True)

if __name__ == "__main__":
    main()
