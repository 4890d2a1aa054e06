#!/usr/bin/env python3
"""Per-wave timeline of the split conv kernel from an instrumented (-DMOTIF_TRACE) build of the library:
   MOTIF_HIP_LIB=tools/_trace/libmotif_hip.so MOTIF_CONV_MMA=6 python tools/trace_split.py [shape index]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from motif_amd import _lib
from motif_amd.models.modules.layers import Conv2d
from tools.conv_bench import SHAPES

n, ci, co, k, s, h, w = SHAPES[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
m = Conv2d(ci, co, k, s, k // 2).cuda()
x = torch.randn(n, ci, h, w, device="cuda")
for _ in range(3):
    y = m(x, act=1)
torch.cuda.synchronize()
lib = _lib.load()
nb = 1024
buf = (ctypes.c_longlong * (nb * 4 * 16))()
lib.motif_debug_trace.restype = ctypes.c_int
rc = lib.motif_debug_trace(buf, nb * 4 * 16)
t = np.frombuffer(buf, dtype=np.int64).reshape(nb, 4, 16).astype(np.float64)
nblk = min(nb, ((h + 7) // 8) * ((w + 31) // 32) * n)
t = t[:nblk]
t0 = t[:, :, 0].min()
# s_memtime ticks at 100 MHz constant clock on gfx9 (REFCLK): report in kcyc
tick_us = 1.0 / 1000  # report kilo-cycles (s_memtime = shader cycles)
names = ["start", "loads issued", "commit0 done", "barrier0", "c0 mfma end", "c0 barrier", "c1 mfma end", "c1 barrier",
         "c2 mfma end", "c2 barrier", "c3 mfma end", "c3 barrier", "-", "-", "epilogue issued", "stores drained"]
d = np.diff(t, axis=2) * tick_us
print("rc", rc, "blocks", nblk, "kernel span %.1f kcyc" % ((t[:, :, 15].max() - t0) * tick_us))
for i in range(15):
    if names[i + 1] == "-" or names[i] == "-":
        continue
    print("%-16s -> %-16s  mean %7.2f kcyc   p10 %7.2f  p90 %7.2f" % (names[i], names[i + 1], d[:, :, i].mean(), np.percentile(d[:, :, i], 10), np.percentile(d[:, :, i], 90)))
print("start -> eoff done      mean %7.2f kcyc" % ((t[:, :, 12] - t[:, :, 0]).mean() * tick_us))
print("eoff done -> W0 issued  mean %7.2f kcyc" % ((t[:, :, 13] - t[:, :, 12]).mean() * tick_us))
print("W0 issued -> loads issued mean %7.2f kcyc" % ((t[:, :, 1] - t[:, :, 13]).mean() * tick_us))
print("c3 barrier -> epilogue issued mean %7.2f kcyc" % ((t[:, :, 14] - t[:, :, 11]).mean() * tick_us))
print("block duration mean %.2f kcyc" % ((t[:, :, 15] - t[:, :, 0]).mean() * tick_us))
st = (t[:, 0, 0] - t0) * tick_us
print("block start times: first-wave blocks (<1us): %d, later: %d; last start %.1f kcyc" % ((st < 1).sum(), (st >= 1).sum(), st.max()))
