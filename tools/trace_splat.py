#!/usr/bin/env python3
"""Phase clocks of splat_owner_kernel<PRE> on the c2 accumulator (2 frames, 720x1280, both directions):
build tools/_trace/libmotif_hip.so with tools/build_trace.sh, then run this on the GPU box."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MOTIF_HIP_LIB", os.path.join(ROOT, "tools", "_trace", "libmotif_hip.so"))
import numpy as np
import torch
from motif_amd import ops

B, N, H, W, s = 1, int(os.environ.get("N", "2")), 180, 320, 4
HH, WW = H * s, W * s
g = torch.Generator().manual_seed(0)
iy = (torch.arange(HH) // s).int().cuda(); ix = (torch.arange(WW) // s).int().cuda()
u = torch.randn(2 * B, 64, HH, WW, generator=g).cuda()
glr = torch.randn(2 * B, 64, H, W, generator=g).cuda()
ab = torch.randn(2, 64, generator=g).cuda()
lo = torch.randn(2 * B * N, 3, H // 4, W // 4, generator=g) * 0.03          # smooth flow, a few pixels
pred = torch.nn.functional.interpolate(lo, size=(HH, WW), mode="bilinear").contiguous().cuda()
alpha = torch.tensor([-20.0]).cuda()
fused = os.environ.get("FUSED", "1") == "1"        # U already holds U + G (what the model does): one load per plane
if fused:
    u = u + glr[:, :, (torch.arange(HH, device="cuda") // s)][:, :, :, (torch.arange(WW, device="cuda") // s)]
def run():
    return ops.splat_motif_pre(u, pred, None if fused else glr, ab, iy, ix, alpha, float(s), B, N, HH, WW, lr_size=(H, W))
for _ in range(3):
    acc = run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    acc = run()
e1.record(); torch.cuda.synchronize()
print("N=%d fused=%d: %.3f ms per call" % (N, fused, e0.elapsed_time(e1) / 5))
lib = ctypes.CDLL(os.environ["MOTIF_HIP_LIB"])
if hasattr(lib, "motif_debug_splat_trace"):
    buf = (ctypes.c_longlong * (2048 * 12))()
    lib.motif_debug_splat_trace(buf, 2048 * 12)
    t = np.array(list(buf), dtype=np.int64).reshape(2048, 12)
    t = t[t[:, 6] > 0]
    k = t.astype(np.float64) / 1000.0
    names = {0: "scan", 1: "list", 2: "scale+buckets", 7: "staging", 3: "barrier", 8: "gather", 4: "barrier", 5: "convert+store", 9: "last gather", 11: "last write"}
    print("  avg kilo-ticks per tile: " + "  ".join("%s %.1f" % (names[i], k[:, i].mean()) for i in (0, 1, 2, 7, 3, 8, 4, 5, 9, 11))
          + "   total %.1f   lifetime %.1f   sources per tile %.0f" % (k[:, [0, 1, 2, 3, 4, 5, 7, 8]].sum(1).mean(), k[:, 10].mean(), t[:, 6].mean()))
    if hasattr(lib, "motif_debug_splat_trace2") and os.environ.get("TIMELINE"):
        buf2 = (ctypes.c_longlong * (64 * 16 * 12))()
        lib.motif_debug_splat_trace2(buf2, 64 * 16 * 12)
        t2 = np.array(list(buf2), dtype=np.int64).reshape(64, 16, 12)
        blk = int(os.environ.get("BLK", "5"))
        base = t2[blk][:, [3, 4, 5, 7, 8]].min()
        print("  chunk 3 of workgroup %d, clocks relative to the first event; columns: staged(7) barrier(3) gathered(8) barrier(4) converted(5)" % blk)
        for w in range(16):
            r = t2[blk][w]
            print("   wave %2d: %7d %7d %7d %7d %7d" % (w, r[7] - base, r[3] - base, r[8] - base, r[4] - base, r[5] - base))
    if os.environ.get("RAW"):
        for r in t[:6]: print("   raw", list(r))
