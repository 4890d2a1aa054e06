#!/usr/bin/env python3
"""Per-phase cycle sums of the split flow_imnet kernel from an instrumented build (tools/build_trace.sh):
   MOTIF_HIP_LIB=tools/_trace/libmotif_hip.so python tools/trace_siren.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from motif_amd import _lib, ops
from motif_amd.models.modules.Ours import LunaTokis, gather_tables
from motif_amd.utils.synth_weights import fill_state_dict

H, W, s, B, N = 180, 320, 4, 1, 3
HH, WW = H * s, W * s
net = fill_state_dict(LunaTokis()).cuda().eval()
iy, ix, ry, rx = gather_tables(H, W, HH, WW, torch.device("cuda"))
feat = torch.randn(2 * B, 64, H, W, device="cuda") * 0.3
times = torch.tensor([[0.0, 0.5, 1.0]], device="cuda")
l0f = ops.conv2d(net.flow_imnet.l0_plan(0, 64), feat)
bf = net.flow_imnet.packed_split(ops.SIREN_FLOW)
for _ in range(2):
    ops.siren_flow(bf, l0f, iy, ix, ry, rx, times, N, HH, WW, pre=2)
torch.cuda.synchronize()
lib = _lib.load()
n = 256 * 8 * 16
buf = (ctypes.c_longlong * n)()
lib.motif_debug_siren_trace.restype = ctypes.c_int
lib.motif_debug_siren_trace(buf, n)
t = np.frombuffer(buf, dtype=np.int64).reshape(256, 8, 16).astype(np.float64)
tiles = 6 * HH * WW / 32 / (256 * 8)
names = ["bookkeeping", "layer0 gather+mfma", "sine1+split", "layer1 mfma", "sine2+split", "L2 chunk mfma (x4)", "chunk sine (x4)", "head partial (x4)", "outputs"]
tot = 0
for i, nm in enumerate(names):
    v = t[:, :, i].mean() / tiles
    tot += v
    print("%-24s %8.0f cycles/tile   (waves 0-3: %8.0f, waves 4-7: %8.0f)" % (nm, v, t[:, :4, i].mean() / tiles, t[:, 4:, i].mean() / tiles))
print("total %.0f cycles/tile/wave, %.1f tiles per wave" % (tot, tiles))
