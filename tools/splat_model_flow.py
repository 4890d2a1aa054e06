#!/usr/bin/env python3
"""The fused PRE splat on flows predicted by flow_imnet with the synthetic weights (what bench.py feeds it): time per call and, with
the trace build (MOTIF_HIP_LIB=tools/_trace/libmotif_hip.so), the distribution of listed sources per tile."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from motif_amd import ops, _lib
from motif_amd.models.modules.Ours import LunaTokis, gather_tables
from motif_amd.utils.synth_weights import fill_state_dict

H, W, s, B, N = 180, 320, 4, 1, int(os.environ.get("N", "2"))
HH, WW = H * s, W * s
net = fill_state_dict(LunaTokis()).cuda().eval()
iy, ix, ry, rx = gather_tables(H, W, HH, WW, torch.device("cuda"))
g = torch.Generator().manual_seed(0)
feat = (torch.randn(2 * B, 64, H, W, generator=g) * 0.3).cuda()
times = torch.tensor([[0.25, 0.75, 0.5][:N]], device="cuda")
pred = ops.siren_flow(net.flow_imnet.packed(), feat, iy, ix, ry, rx, times, N, HH, WW)
print("pred: |p0| mean %.4f max %.4f   flow px: mean %.2f max %.2f" % (pred[:, 0].abs().mean(), pred[:, 0].abs().max(), pred[:, 0].abs().mean() * 80, pred[:, 0].abs().max() * 80))
u = torch.randn(2 * B, 64, HH, WW, generator=g).cuda()
ab = torch.randn(2, 64, generator=g).cuda()
def run():
    return ops.splat_motif_pre(u, pred, None, ab, iy, ix, net.alpha, float(s), B, N, HH, WW, lr_size=(H, W))
for _ in range(3): acc = run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): acc = run()
e1.record(); torch.cuda.synchronize()
print("N=%d: %.3f ms per call" % (N, e0.elapsed_time(e1) / 5))
lib = _lib.load()
if hasattr(lib, "motif_debug_splat_trace"):
    buf = (ctypes.c_longlong * (2048 * 12))()
    lib.motif_debug_splat_trace(buf, 2048 * 12)
    t = np.array(list(buf), dtype=np.int64).reshape(2048, 12)
    c = t[:, 6]; c = c[c > 0]
    print("sources per tile: mean %.0f  p50 %d  p90 %d  p99 %d  max %d   tiles over 2304: %.1f %%  over 2560: %.1f %%  over 3072: %.1f %%"
          % (c.mean(), np.percentile(c, 50), np.percentile(c, 90), np.percentile(c, 99), c.max(), 100.0 * (c > 2304).mean(), 100.0 * (c > 2560).mean(), 100.0 * (c > 3072).mean()))
