#!/usr/bin/env python3
"""BASELINE config 3 (Vimeo-7 septuplet shape: 7 LR frames 256x448, x4 spatial, x8 temporal = 9 timestamps):
output agreement and speed of the arithmetic modes (fp32 MFMA, bf16x3 split, f16x2 split, bf16x2, plain bf16 convolutions)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from motif_amd import ops
from motif_amd.data.synthetic import synthetic_sample
from motif_amd.models import create_model
from motif_amd.option import default_opt
from motif_amd.utils.synth_weights import fill_state_dict


def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 99.0 if mse == 0 else 10 * np.log10(1.0 / mse)


def main():
    h, w, s, T, n = 256, 448, 4, 9, 7
    if len(sys.argv) > 2:
        h, w = int(sys.argv[1]), int(sys.argv[2])
    model = create_model(default_opt(scale=s, gpu_ids=[0]))
    fill_state_dict(model.netG)
    smp = synthetic_sample(h, w, s, T, n_frames=n) if "n_frames" in synthetic_sample.__code__.co_varnames else synthetic_sample(h, w, s, T)
    data = {"LQs": smp["LQs"].cuda(), "GT": smp["GT"][:, :1].cuda(), "time": [t.cuda() for t in smp["time"]], "scale": smp["scale"]}
    print("LQs", tuple(data["LQs"].shape), "timestamps", len(data["time"]))
    outs = {}
    for mode in ("fp32", "bf16x3", "f16x2", "bf16x2", "bf16"):
        ops.set_mma(mode)
        for _ in range(2):
            model.feed_data(data); model.test()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            model.feed_data(data); model.test()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        outs[mode] = model.fake_H.clone()
        px = T * h * s * w * s
        print("%-7s %7.1f ms/clip  %6.1f M HR px/s" % (mode, dt * 1e3, px / dt / 1e6), end="")
        if mode != "fp32":
            d = outs[mode] - outs["fp32"]
            print("   PSNR vs fp32 %.1f dB  Linf %.2e" % (psnr(outs[mode], outs["fp32"]), float(d.abs().max())))
        else:
            print()


if __name__ == "__main__":
    main()
