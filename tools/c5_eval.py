#!/usr/bin/env python3
"""BASELINE config 5 (one 540x960 LR 4-frame clip -> 2160x3840, x4 spatial, x4 temporal = 5 timestamps) on ONE GPU:
untiled render vs the row-band tile mode the 8-GPU job uses (8 bands rendered one after another here), agreement and
per-stage times -> projected 8-GPU time = replicated LR stage + slowest band."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motif_amd import dist as md
from motif_amd.data.synthetic import synthetic_sample
from motif_amd.models.modules.Ours import LunaTokis
from motif_amd.utils.synth_weights import fill_state_dict


def sync_time(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, time.perf_counter() - t0


def main():
    h, w, s, T, bands, halo = 540, 960, 4, 5, 8, 64
    if len(sys.argv) > 2:
        h, w = int(sys.argv[1]), int(sys.argv[2])
    HH, WW = h * s, w * s
    net = fill_state_dict(LunaTokis()).cuda().eval()
    smp = synthetic_sample(h, w, s, T)
    x = smp["LQs"].cuda(); times = [t.cuda() for t in smp["time"]]; scale = smp["scale"]

    def untiled():
        net.clear_cache(); outs = []
        with torch.no_grad():
            for l in range(0, T, 3):
                outs.append(net(x, None, times[l:l + 3], scale, use_GT=False, iter=4)[0])
        return torch.cat(outs, 0)

    untiled()
    full, t_full = sync_time(untiled)
    print("untiled: %.1f ms per clip, %.1f M HR px/s" % (t_full * 1e3, T * HH * WW / t_full / 1e6))

    # tile mode: LR stage once (it is replicated on every rank), then each band
    def lr_stage():
        net.clear_cache(); net.band = (0, 8)
        with torch.no_grad():
            net._cache, net._cache_key = net._clip_stage(x, HH, WW, 4), None
    _, t_lr = sync_time(lr_stage)
    _, t_lr = sync_time(lr_stage)
    parts, t_bands, worst = [], [], 0.0
    for r in range(bands):
        net.band, net.band_halo = md.band_of(HH, r, bands, 8), halo
        def band():
            outs = []
            with torch.no_grad():
                for l in range(0, T, 3):
                    outs.append(net(x, None, times[l:l + 3], scale, use_GT=False, iter=4)[0])
            return torch.cat(outs, 0)
        key = (x.data_ptr(), x._version, tuple(x.shape), HH, WW, 4, False)
        net._cache_key = key; net._cache["x"] = x
        o, tb = sync_time(band)
        parts.append(o); t_bands.append(tb); worst = max(worst, float(net.last_max_flow_y))
    net.band = None
    tiled = torch.cat(parts, dim=-2)
    d = (tiled - full).abs()
    print("tile mode (8 row bands, halo %d): max |flow_y| %.1f px; max |tiled - untiled| = %.2e" % (halo, worst, float(d.max())))
    print("LR stage %.1f ms (replicated); bands %s ms" % (t_lr * 1e3, " ".join("%.1f" % (t * 1e3) for t in t_bands)))
    t8 = t_lr + max(t_bands)
    print("projected 8-GPU clip time %.1f ms = %.1f M HR px/s (1 GPU untiled: %.1f ms); HR-stage overhead of the halo: %.2fx"
          % (t8 * 1e3, T * HH * WW / t8 / 1e6, t_full * 1e3, sum(t_bands) / max(t_full - t_lr, 1e-9)))


if __name__ == "__main__":
    main()
