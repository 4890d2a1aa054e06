#!/usr/bin/env python3
"""Which host lines launch the torch glue of a clip: torch.profiler with stacks over one warm clip of the bench workload.
Part 1: aten ops that own device time (copies, cat, index_select, fill ...), grouped by (op, innermost motif_amd frame).
Part 2: every device-side memcpy / memset activity (the runtime's blit kernels: __amd_rocclr_copyBuffer in a rocprofv3 trace) with the
CPU op and python frame it was issued under -- host-to-device transfers of small host tensors show up here, not as aten kernels."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    model.feed_data(data); model.test()
    torch.cuda.synchronize()


def frame_of(ev):
    for fr in ev.stack or []:
        if "motif_amd" in fr and "ops.py" not in fr:
            return fr.split("motif_amd/")[-1]
    for fr in ev.stack or []:
        if "motif_amd" in fr:
            return fr.split("motif_amd/")[-1]
    return "?"


agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0:
        continue
    if ev.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue                                         # count the innermost op that owns the kernel
    a = agg[(ev.name, frame_of(ev))]
    a[0] += 1; a[1] += ev.device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print("torch ops with device time in one clip: %d launches, %.3f ms" % (sum(v[0] for _, v in rows), tot / 1e3))
for (name, where), (n, us) in rows[:45]:
    print("%8.1f us %4d  %-28s %s" % (us, n, name, where))

print("\ndevice-side memcpy / memset activities and non-aten runtime calls that issue them:")
mem = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    nm = ev.name
    if not any(k in nm for k in ("Memcpy", "Memset", "hipMemcpy", "hipMemset", "copyBuffer", "fillBuffer")):
        continue
    parent, where = "?", "?"
    p = ev.cpu_parent
    while p is not None:
        if parent == "?" and (p.name.startswith("aten::") or "motif" in p.name):
            parent = p.name
        if where == "?" and frame_of(p) != "?":
            where = frame_of(p)
        p = p.cpu_parent
    if where == "?":
        where = frame_of(ev)
    m = mem[(nm[:60], parent, where)]
    m[0] += 1; m[1] += max(ev.device_time_total, 0)
for (nm, parent, where), (n, us) in sorted(mem.items(), key=lambda kv: -kv[1][0])[:60]:
    print("%5d  %8.1f us  %-44s under %-22s %s" % (n, us, nm, parent, where))
