"""Residual-block chains: ops.resblock_chain as ONE launch (motif_conv2d_chain_fwd) against the launches layer by layer, same box.
   python tools/chain_time.py   (MOTIF_HIP_LIB=tools/_trace/<variant>.so for a variant build: CHAIN_DEFER / CHAIN_ABL in conv_wino.hip)"""
import sys, torch
sys.path.insert(0, ".")
from motif_amd import ops
torch.manual_seed(0)
dev = "cuda"
ops.CONV_CHAIN_MIN_TILES = 1
def mk(c, nb):
    return [tuple(ops.ConvPlan(torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5), torch.randn(c, device=dev) * 0.1, 1, 1, 1, 1, 0) for _ in range(2)) for _ in range(nb)]
for (n, h, w, nb) in ((3, 180, 320, 40), (6, 180, 320, 40), (9, 180, 320, 40), (2, 180, 320, 5), (1, 90, 160, 40)):
    blocks = mk(64, nb); x = torch.randn(n, 64, h, w, device=dev)
    res = []
    for flag in (False, True):
        ops.CONV_CHAIN = flag
        for _ in range(3): ops.resblock_chain(blocks, x)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(4):
            e0.record()
            for _ in range(5): ops.resblock_chain(blocks, x)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        res.append(best)
    print("N %d %dx%d blocks %d: per-layer %.3f ms  chain %.3f ms" % (n, h, w, nb, res[0], res[1]), flush=True)
