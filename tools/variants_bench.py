#!/usr/bin/env python3
"""Clip times of the three generators at BASELINE config-2 size (4-frame 180x320 -> 720x1280, 7 timestamps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motif_amd.data.synthetic import synthetic_sample
from motif_amd.models import create_model
from motif_amd.option import default_opt
from motif_amd.utils.synth_weights import fill_state_dict

for which in ("Ours", "Ours_4", "Ours_44"):
    model = create_model(default_opt(scale=4, gpu_ids=[0], which_model_G=which))
    fill_state_dict(model.netG)
    s = synthetic_sample(180, 320, 4, 7)
    data = {"LQs": s["LQs"].cuda(), "GT": s["GT"][:, :1].cuda(), "time": [t.cuda() for t in s["time"]]}
    if which != "Ours_44":
        data["scale"] = s["scale"]
    for _ in range(2):
        model.feed_data(data); model.test()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        model.feed_data(data); model.test()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    assert model.fake_H.shape == (7, 1, 3, 720, 1280) and torch.isfinite(model.fake_H).all()
    print("%-8s %6.1f ms per clip  %6.1f M HR px/s" % (which, dt * 1e3, 7 * 720 * 1280 / dt / 1e6))
