#!/usr/bin/env python3
"""Where do the device-to-device copies / cats of one clip come from?  (torch profiler, python stacks)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

def main():
    from motif_amd.data.synthetic import synthetic_sample
    from motif_amd.models import create_model
    from motif_amd.option import default_opt
    from motif_amd.utils.synth_weights import fill_state_dict
    model = create_model(default_opt(scale=4, gpu_ids=[0]))
    fill_state_dict(model.netG)
    s = synthetic_sample(180, 320, 4, 7, seed=0)
    data = {"LQs": s["LQs"].cuda(), "GT": s["GT"][:, :1].cuda(), "time": [t.cuda() for t in s["time"]], "scale": s["scale"]}
    for _ in range(2):
        model.feed_data(data); model.test()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
        model.feed_data(data); model.test()
        torch.cuda.synchronize()
    cnt = collections.Counter()
    for ev in prof.events():
        if ev.name in ("aten::copy_", "aten::_to_copy", "aten::cat", "aten::clone", "aten::stack", "aten::repeat", "aten::flip", "aten::zeros", "aten::fill_", "aten::zero_", "aten::arange", "aten::full", "aten::lift_fresh", "hipMemcpyAsync", "hipMemcpyWithStream", "hipMemcpy"):
            cnt[(ev.name, str(ev.input_shapes)[:60])] += 1
    for (name, where), n in cnt.most_common(60):
        print("%4d  %-18s %s" % (n, name, where))

if __name__ == "__main__":
    main()
