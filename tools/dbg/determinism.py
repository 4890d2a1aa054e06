import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from motif_amd import ops
from motif_amd.data.synthetic import synthetic_sample
from motif_amd.models.modules.Ours import LunaTokis
from motif_amd.utils.synth_weights import fill_state_dict
ops.set_mma(sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
net = fill_state_dict(LunaTokis()).cuda().eval()
import os
net.overlap_raft = not os.environ.get("NO_OVERLAP")
s = synthetic_sample(180, 320, 4, 7)
x = s["LQs"].cuda(); times = [t.cuda() for t in s["time"]]
runs = []
with torch.no_grad():
    for r in range(3):
        net.clear_cache(); st = {}
        o, f, _ = net(x, None, times[6:7], s["scale"], use_GT=False, iter=4, stages=st)
        runs.append({k: v.clone() for k, v in st.items() if torch.is_tensor(v)} | {"out": o.clone()})
for k in runs[0]:
    d1 = float((runs[1][k] - runs[0][k]).abs().max()); d2 = float((runs[2][k] - runs[0][k]).abs().max())
    print("%-14s run1-run0 %.3e   run2-run0 %.3e" % (k, d1, d2))
d = (runs[1]["feat"] - runs[0]["feat"]).abs()
print("feat shape", tuple(d.shape), "n diff", int((d > 0).sum()), "of", d.numel())
nz = torch.nonzero(d > 1e-6)
if nz.numel():
    for dim in range(nz.shape[1]):
        u = torch.unique(nz[:, dim])
        print(" dim", dim, "min", int(u.min()), "max", int(u.max()), "count", u.numel())
