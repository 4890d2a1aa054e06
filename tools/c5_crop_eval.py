#!/usr/bin/env python3
"""BASELINE config 5 in CROPPED tile mode (motif_amd.dist.render_clip_tiled(lr_halo=R)) on ONE GPU: the 8 ranks' crops are
rendered one after another; PSNR against the untiled render and per-rank times -> projected 8-GPU clip time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from motif_amd import dist as md
from motif_amd.data.synthetic import synthetic_sample
from motif_amd.models.modules.Ours import LunaTokis
from motif_amd.utils.synth_weights import fill_state_dict


def main():
    h, w, s, T, bands, halo = 540, 960, 4, 5, 8, 64
    HH, WW = h * s, w * s
    net = fill_state_dict(LunaTokis()).cuda().eval()
    smp = synthetic_sample(h, w, s, T)
    x = smp["LQs"].cuda(); times = [t.cuda() for t in smp["time"]]

    def render(xr, sc):
        outs = []
        with torch.no_grad():
            for l in range(0, T, 3):
                outs.append(net(xr, None, times[l:l + 3], sc, use_GT=False, iter=4)[0])
        return torch.cat(outs, 0)

    render(x, smp["scale"]); net.clear_cache()
    torch.cuda.synchronize(); t0 = time.perf_counter(); full = render(x, smp["scale"]); torch.cuda.synchronize()
    t_full = time.perf_counter() - t0
    print("untiled: %.1f ms per clip" % (t_full * 1e3))
    for R in [int(v) for v in os.environ.get("LR_HALOS", "16,32,48,64").split(",")]:
        parts, ts = [], []
        for r in range(bands):
            band = md.band_of(HH, r, bands, 16)
            a, b = md.crop_rows_for_band(band, halo, h, HH, R)
            xr = x[..., a:b, :].contiguous()
            net.band, net.band_halo = (band[0] - a * s, band[1] - a * s), halo
            net.clear_cache()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            parts.append(render(xr, [[(b - a) * s], [WW]]))
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        net.band = None
        tiled = torch.cat(parts, dim=-2)
        mse = float(((tiled.double() - full.double()) ** 2).mean())
        psnr = 99.0 if mse == 0 else 10 * np.log10(1.0 / mse)
        print("lr_halo %3d: PSNR(cropped tiles, untiled) = %.2f dB, Linf %.2e; per-rank ms %s -> 8-GPU clip %.1f ms = %.2fx" % (
            R, psnr, float((tiled - full).abs().max()), " ".join("%.0f" % (t * 1e3) for t in ts), max(ts) * 1e3, t_full / max(ts)))


if __name__ == "__main__":
    main()
