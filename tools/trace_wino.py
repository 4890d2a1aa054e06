#!/usr/bin/env python3
"""Per-wave timeline of conv_wino_kernel from an instrumented (-DMOTIF_TRACE) build of the library:
   MOTIF_HIP_LIB=tools/_trace/libmotif_hip.so python tools/trace_s2.py [shape index]
   slots: 0 start, 1 prologue done (first chunk staged), then one per chunk (after its barrier) and one per tile epilogue, 31 end"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from motif_amd import _lib, ops
from motif_amd.models.modules.layers import Conv2d
from tools.conv_bench import SHAPES

n, ci, co, k, s, h, w = SHAPES[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
m = Conv2d(ci, co, k, s, k // 2).cuda()
x = torch.randn(n, ci, h, w, device="cuda")
res = torch.randn(n, co, h // s, w // s, device="cuda") if os.environ.get("RES") else None
kw = dict(act=int(os.environ.get("ACT", "1")), res=res, res_mode=1) if res is not None else dict(act=int(os.environ.get("ACT", "1")))
for _ in range(3):
    y = m(x, **kw)
torch.cuda.synchronize()
if os.environ.get("MMA"):
    ops.set_conv_mma(int(os.environ["MMA"]))
ops.set_option("conv_engine", 5)
if os.environ.get("CHAIN"):     # CHAIN=<blocks>: the traced launch is ONE chain launch of that many residual blocks (ops.resblock_chain)
    nbk = int(os.environ["CHAIN"])
    blocks = [tuple(ops.ConvPlan(torch.randn(co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5), torch.randn(co, device="cuda") * 0.1, 1, 1, 1, 1, 0) for _ in range(2)) for _ in range(nbk)]
    for _ in range(int(os.environ.get("REPS", "2"))):
        y = ops.resblock_chain(blocks, x)
else:
  for _ in range(int(os.environ.get("REPS", "2"))):      # REPS=500: the traced (last) launch sees the clock of a sustained run
    y = m(x, **kw)
torch.cuda.synchronize()
lib = _lib.load()
nb = 1024
buf = (ctypes.c_longlong * (nb * 4 * 32))()
lib.motif_debug_wino_trace.restype = ctypes.c_int
rc = lib.motif_debug_wino_trace(buf, nb * 4 * 32)
t = np.frombuffer(buf, dtype=np.int64).reshape(nb, 4, 32).astype(np.float64)[:256]
ok = t[:, :, 1] > 0
nch = (ci + 15) // 16
print("rc", rc, "shape", (n, ci, co, h, w), "blocks traced", int(ok[:, 0].sum()), "chunks per tile", nch)
print("prologue %.2f kcyc" % ((t[:, :, 1] - t[:, :, 0])[ok].mean() / 1e3))
lab = ["body->end", "epilogue", "barrier"]
for i in range(9):
    row = []
    for k3 in range(3):
        sl = 2 + 3 * i + k3
        v = ok & (t[:, :, sl] > 0) & (t[:, :, sl - 1] > 0)
        row.append("%s %6.2f" % (lab[k3] if k3 else "chunk(from prev stamp)", ((t[:, :, sl] - t[:, :, sl - 1])[v].mean() / 1e3) if v.any() else float("nan")))
    print("  iteration %d (chunk %d): %s" % (i, i % nch, " | ".join(row)))
print("block duration mean %.2f kcyc, max %.2f" % ((t[:, :, 31] - t[:, :, 0])[ok].mean() / 1e3, (t[:, :, 31] - t[:, :, 0])[ok].max() / 1e3))
rt = (t[:, :, 29] - t[:, :, 30])[ok] / 100.0                      # us on the constant 100 MHz counter
print("block duration mean %.2f us -> shader clock %.0f MHz while the kernel runs" % (rt.mean(), ((t[:, :, 31] - t[:, :, 0])[ok] / rt).mean()))

buf2 = (ctypes.c_longlong * (256 * 4 * 8 * 8))()
lib.motif_debug_wino_trace2.restype = ctypes.c_int
rc = lib.motif_debug_wino_trace2(buf2, 256 * 4 * 8 * 8)
t2 = np.frombuffer(buf2, dtype=np.int64).reshape(256, 4, 8, 8).astype(np.float64)
print("epilogue of the first tiles (cycles): setup + residual requests | transform + transpose of pass 0 | pass 0 | pass 1 | pass 2 | pass 3")
for k in range(3):
    okc = (t2[:, :, k, 0] > 0) & (t2[:, :, k, 6] > 0)
    if okc.any():
        d = t2[:, :, k, 1:7] - t2[:, :, k, 0:6]
        print("  tile %d: %s | total %6.0f" % (k, " ".join("%6.0f" % d[..., i][okc].mean() for i in range(6)), (t2[:, :, k, 6] - t2[:, :, k, 0])[okc].mean()))
if os.environ.get("SS"):      # a -DMOTIF_TRACE_SS build: entries 3 .. 7 hold the per-super-step stamps of chunks 3 .. 7 of the workgroup
    print("super-step durations (cycles) of chunks 3 .. 7: ss0 .. ss5")
    for k in range(3, 8):
        okc = (t2[:, :, k, 0] > 0) & (t2[:, :, k, 6] > 0)
        if okc.any():
            d = t2[:, :, k, 1:7] - t2[:, :, k, 0:6]
            print("  chunk %d (chunk %d of its tile): %s | total %6.0f" % (k, k % nch, " ".join("%6.0f" % d[..., i][okc].mean() for i in range(6)), (t2[:, :, k, 6] - t2[:, :, k, 0])[okc].mean()))
            if k == 5:
                print("     per wave: " + " | ".join(" ".join("%5.0f" % d[:, w, i][okc[:, w]].mean() for i in range(6)) for w in range(4)))

if os.environ.get("PRO"):
    d = t2[:, :, 7, 1:6] - t2[:, :, 7, 0:5]
    okc = t2[:, :, 7, 5] > 0
    print("prologue phases (cycles): table+barrier | issue weights+row pieces | wait for them | staging pieces | init + step-1 requests + barrier")
    print("   " + " ".join("%6.0f" % d[..., i][okc].mean() for i in range(5)))
    print("   per wave: " + " | ".join(" ".join("%5.0f" % d[:, w, i][okc[:, w]].mean() for i in range(5)) for w in range(4)))

if os.environ.get("CHAIN"):
    buf3 = (ctypes.c_longlong * (256 * 8 * 8))()
    lib.motif_debug_chain_trace.restype = ctypes.c_int
    lib.motif_debug_chain_trace(buf3, 256 * 8 * 8)
    t3 = np.frombuffer(buf3, dtype=np.int64).reshape(256, 8, 8).astype(np.float64)
    print("chain, wave 0, end of a tile (cycles): init_acc | ticket take | entry | issue 2 atomics | store wait | barrier | publish")
    for k in range(1, 5):
        okc = (t3[:, k, 0] > 0) & (t3[:, k, 7] > 0) & (t3[:, k, 2] > 0)
        if okc.any():
            d = t3[:, k, 1:8] - t3[:, k, 0:7]
            print("  tile %d: %s" % (k, " ".join("%6.0f" % d[:, i][okc].mean() for i in range(7))))
