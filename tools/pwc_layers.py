#!/usr/bin/env python3
"""Per-launch timing of one PWC-Net forward on a 720x1280 pair (event pairs around every ops.* call, serialised): which layers carry
the time, and on which engine (ISA kernel names come from tools/prof_pwc.sh)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motif_amd import ops
from motif_amd.OpticalFlow.PWCNet import PWCNet
from motif_amd.utils.synth_weights import fill_state_dict

net = fill_state_dict(PWCNet()).cuda().eval()
f0, f1 = torch.rand(1, 3, 720, 1280, device="cuda"), torch.rand(1, 3, 720, 1280, device="cuda")
rec = []


def wrap(name):
    orig = getattr(ops, name)

    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(*a, **k)
        e1.record()
        desc = ""
        if name == "conv2d":
            plan, x = a[0], a[1]
            w = plan.weight
            desc = "x%s w%s s%d d%d" % (tuple(x.shape), tuple(w.shape), plan.stride, plan.dil)
        else:
            desc = " ".join(str(tuple(t.shape)) for t in a if torch.is_tensor(t))[:80]
        rec.append((name, desc, e0, e1))
        return out
    setattr(ops, name, f)


for n in ("conv2d", "corr81", "deconv4x4s2", "pwc_backward_warp", "resize_bilinear", "axpby"):
    wrap(n)
with torch.no_grad():
    for _ in range(2):
        rec.clear()
        net(f0, f1)
        torch.cuda.synchronize()
tot = 0.0
agg = collections.OrderedDict()
for name, desc, e0, e1 in rec:
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    a = agg.setdefault((name, desc), [0, 0.0])
    a[0] += 1; a[1] += us
print("one pair: %d launches, %.1f us inside the event pairs" % (len(rec), tot))
for (name, desc), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%8.1f us %3d  %-18s %s" % (us, n, name, desc))
