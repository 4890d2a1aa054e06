#!/usr/bin/env python3
"""Full-size parity of the HIP path against the CPU oracle: one clip of BASELINE config 2 (LR 180x320 -> 720x1280; default) or, with
CONFIG=c3, config 3 (Vimeo-7 septuplet shape: 7 LR frames 256x448 -> 1024x1792).  TIMES = the timestamps rendered (default one: 0.5),
MODES = the arithmetics.  Slow on the CPU side (minutes); run by hand, result recorded in DESIGN.md / profiles/."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from motif_amd.data.synthetic import synthetic_sample
from motif_amd.models.modules.Ours import LunaTokis
from motif_amd.utils.synth_weights import fill_state_dict
from motif_amd.utils import util
from oracle.motif_ref import MotifRef

def main():
    times = [float(t) for t in os.environ.get("TIMES", "0.5").split(",")]
    c3 = os.environ.get("CONFIG", "c2") == "c3"
    s = synthetic_sample(256, 448, 4, 9, n_frames=7) if c3 else synthetic_sample(180, 320, 4, 7)
    tl = [torch.full((1, 1), t) for t in times]
    from motif_amd import ops
    net = fill_state_dict(LunaTokis()).cuda().eval()
    outs = {}
    for mode in os.environ.get("MODES", "f16x2,bf16x3,fp32").split(","):
        ops.set_mma(mode)
        net.clear_cache()
        with torch.no_grad():
            outs[mode] = net(s["LQs"].cuda(), None, [t.cuda() for t in tl], s["scale"], use_GT=False, iter=4)[:2]
        torch.cuda.synchronize()
    t0 = time.time()
    with torch.no_grad():
        ref, rflow, _ = fill_state_dict(MotifRef().eval())(s["LQs"], None, tl, s["scale"], use_GT=False, iter=4)
    dt = time.time() - t0
    gt = s["GT"][0, 1:1 + len(times)]
    pr = util.y_psnr_per_frame(gt, ref[:, 0])
    print("%s full size (LR %s), %d timestamp(s): oracle %.1f s on %d threads" % ("c3" if c3 else "c2", tuple(s["LQs"].shape), len(times), dt, torch.get_num_threads()))
    for mode, (out, flow) in outs.items():
        o = out.cpu()
        mse = float(((o.double() - ref.double()) ** 2).mean())
        pm = util.y_psnr_per_frame(gt, o[:, 0])
        print("[%s] PSNR(build, oracle) = %.2f dB   Linf = %.3e   flow Linf = %.3e px(LR units)   Y-PSNR vs GT |delta| max %.5f dB"
              % (mode, 10 * np.log10(1.0 / mse), float((o - ref).abs().max()), float((flow.cpu() - rflow).abs().max()), float(np.abs(pm - pr).max())))

if __name__ == "__main__":
    main()
