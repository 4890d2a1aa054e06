#!/usr/bin/env python3
"""Per-wave phase timeline of the ping-pong conv kernel from an instrumented (-DMOTIF_TRACE) build of the library:
   MOTIF_HIP_LIB=tools/_trace/libmotif_hip.so python tools/trace_pp.py [shape index] [rp]
   slots: 0 start | per phase-loop iteration k: 1+4k other done, 2+4k barrier passed, 3+4k compute done, (next 1+4k..) | 31 end"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from motif_amd import _lib, ops
from motif_amd.models.modules.layers import Conv2d
from tools.conv_bench import SHAPES

n, ci, co, k, s, h, w = SHAPES[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
if len(sys.argv) > 2:
    ops.set_option("pp_rp", int(sys.argv[2]))
m = Conv2d(ci, co, k, s, k // 2).cuda()
x = torch.randn(n, ci, h, w, device="cuda")
res = torch.randn(n, co, h // s, w // s, device="cuda") if os.environ.get("RES") else None
kw = dict(act=1, res=res, res_mode=1) if res is not None else dict(act=1)
for _ in range(3):
    y = m(x, **kw)
torch.cuda.synchronize()
lib = _lib.load()
nb = 1024
buf = (ctypes.c_longlong * (nb * 8 * 32))()
lib.motif_debug_pp_trace.restype = ctypes.c_int
rc = lib.motif_debug_pp_trace(buf, nb * 8 * 32)
t = np.frombuffer(buf, dtype=np.int64).reshape(nb, 8, 32).astype(np.float64)
t = t[:256]
ok = t[:, :, 0] > 0
t0 = t[:, :, 0][ok].min()
print("rc", rc, "shape", (n, ci, co, h, w), "blocks traced", int(ok[:, 0].sum()))
print("kernel span %.1f kcyc" % ((t[:, :, 31].max() - t0) / 1e3))
for hh in (0, 1):
    tw = t[:, 4 * hh:4 * hh + 4]
    print("half %d:" % hh)
    for kk in range(5):
        a, b, c, d = 4 * kk, 1 + 4 * kk, 2 + 4 * kk, 3 + 4 * kk
        if d > 20:
            break
        prev = tw[:, :, a] if kk else tw[:, :, 0]
        other = (tw[:, :, b] - prev)
        barw = (tw[:, :, c] - tw[:, :, b])
        comp = (tw[:, :, d] - tw[:, :, c])
        nxt = (tw[:, :, d + 1] - tw[:, :, d]) if d + 1 < 31 else comp * 0
        v = tw[:, :, d] > 0
        if not v.any():
            break
        print("  k=%d  other %6.2f  wait-barrier %6.2f  compute %6.2f  (kcyc, mean over waves; compute p10 %5.2f p90 %5.2f)" % (
            kk, other[v].mean() / 1e3, barw[v].mean() / 1e3, comp[v].mean() / 1e3, np.percentile(comp[v], 10) / 1e3, np.percentile(comp[v], 90) / 1e3))
for hh in (0, 1):
    tw = t[:, 4 * hh:4 * hh + 4]
    v = tw[:, :, 29] > 0
    if v.any():
        d = lambda a, b: (tw[:, :, b] - tw[:, :, a])[v].mean() / 1e3
        print("half %d first tile boundary: epilogue %.2f  zero+bias %.2f  commit+w0 %.2f kcyc (requests issued -> ...)" % (hh, d(26, 27), d(27, 28), d(28, 29)))
for hh in (0, 1):
    tw = t[:, 4 * hh:4 * hh + 4]
    v = tw[:, :, 24] > 0
    if v.any():
        d = lambda a, b: (tw[:, :, b] - tw[:, :, a])[v].mean() / 1e3
        print("half %d phase k=1: weights+DMA issued %.2f (from phase start)  DMA landed +%.2f  split +%.2f kcyc" % (hh, d(4, 22), d(22, 23), d(23, 24)))
print("block duration mean %.2f kcyc, max %.2f" % ((t[:, :, 31] - t[:, :, 0])[ok].mean() / 1e3, (t[:, :, 31] - t[:, :, 0])[ok].max() / 1e3))
