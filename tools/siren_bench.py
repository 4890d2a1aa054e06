#!/usr/bin/env python3
"""Time the three SIREN kernels (and the fused splat) in isolation at c2 size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motif_amd import ops
from motif_amd.models.modules.Ours import LunaTokis, gather_tables
from motif_amd.utils.synth_weights import fill_state_dict

def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def main():
    H, W, s, B, N = 180, 320, 4, 1, 3
    HH, WW = H * s, W * s
    Q = HH * WW
    net = fill_state_dict(LunaTokis()).cuda().eval()
    iy, ix, ry, rx = gather_tables(H, W, HH, WW, torch.device("cuda"))
    feat = torch.randn(2 * B, 64, H, W, device="cuda") * 0.3
    times = torch.tensor([[0.0, 0.5, 1.0]], device="cuda")
    reps = int(os.environ.get("REPS", "5"))
    which = os.environ.get("ONLY", "flow,synth,imnet,splat").split(",")
    pred = ops.siren_flow(net.flow_imnet.packed(), feat, iy, ix, ry, rx, times, N, HH, WW)
    imn = ops.siren_imnet(net.imnet.packed(), feat, iy, ix, ry, rx, HH, WW)
    acc = ops.splat_motif(imn, pred, feat, iy, ix, net.alpha, HH / H, B, N, HH, WW)
    res = feat[:B].contiguous()
    if "flow" in which:
        ms = t(lambda: ops.siren_flow(net.flow_imnet.packed(), feat, iy, ix, ry, rx, times, N, HH, WW), reps)
        print("flow_imnet N=3: %.3f ms  %.1f TFLOP/s" % (ms, 2 * 25536 * 2 * B * N * Q / ms / 1e9))
    if "synth" in which:
        ms = t(lambda: ops.siren_synth(net.synth_net.packed(), acc, res, iy, ix, times, B, N, HH, WW), reps)
        print("synth N=3: %.3f ms  %.1f TFLOP/s" % (ms, 2 * 38016 * B * N * Q / ms / 1e9))
    if "imnet" in which:
        ms = t(lambda: ops.siren_imnet(net.imnet.packed(), feat, iy, ix, ry, rx, HH, WW), reps)
        print("imnet: %.3f ms  %.1f TFLOP/s" % (ms, 2 * 41088 * 2 * B * Q / ms / 1e9))
    if "split" in which:
        l0i = ops.conv2d(net.imnet.l0_plan(0, 64), feat)
        l0f = ops.conv2d(net.flow_imnet.l0_plan(0, 64), feat)
        l0s = ops.conv2d(net.synth_net.l0_plan(133, 197), res)
        for pre, name in ((1, "fp32 pre"), (2, "three bf16 parts"), (3, "two fp16 parts")):
            pk = (lambda kind, m: ops.siren_pack_split(kind, m.linears(), pre=pre)) if pre >= 2 else (lambda kind, m: m.packed())
            bf, bs, bi = pk(ops.SIREN_FLOW, net.flow_imnet), pk(ops.SIREN_SYNTH, net.synth_net), pk(ops.SIREN_IMNET, net.imnet)
            ms = t(lambda: ops.siren_flow(bf, l0f, iy, ix, ry, rx, times, N, HH, WW, pre=pre), reps)
            print("[%s] flow_imnet N=3: %.3f ms  %.1f TFLOP/s" % (name, ms, 2 * 25536 * 2 * B * N * Q / ms / 1e9))
            ms = t(lambda: ops.siren_synth(bs, acc, l0s, iy, ix, times, B, N, HH, WW, pre=pre), reps)
            print("[%s] synth N=3: %.3f ms  %.1f TFLOP/s" % (name, ms, 2 * 38016 * B * N * Q / ms / 1e9))
            ms = t(lambda: ops.siren_imnet(bi, l0i, iy, ix, ry, rx, HH, WW, pre=pre), reps)
            print("[%s] imnet: %.3f ms  %.1f TFLOP/s" % (name, ms, 2 * 41088 * 2 * B * Q / ms / 1e9))
    if "splat" in which:
        ms = t(lambda: ops.splat_motif(imn, pred, feat, iy, ix, net.alpha, HH / H, B, N, HH, WW, acc=acc), reps)
        print("splat N=3: %.3f ms  %.2f TB/s algorithmic (2640 B/px-frame)" % (ms, 2640.0 * B * N * Q / ms / 1e9))

if __name__ == "__main__":
    main()
