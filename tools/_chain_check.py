import sys, time, torch
sys.path.insert(0, ".")
from motif_amd import ops
torch.manual_seed(0)
dev = "cuda"
def mk(c, nb):
    blocks = []
    for _ in range(nb):
        ps = []
        for _ in range(2):
            w = (torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5)).requires_grad_(False)
            b = torch.randn(c, device=dev) * 0.1
            ps.append(ops.ConvPlan(w, b, 1, 1, 1, 1, 0))
        blocks.append(tuple(ps))
    return blocks
def run(n, c, h, w, nb, strided=False, reps=3):
    blocks = mk(c, nb)
    x = torch.randn(n, c, h, w, device=dev)
    def mkout():
        if strided:
            big = torch.zeros(n, 2, c, h, w, device=dev)
            return big[:, 0]
        return None
    ops.CONV_CHAIN = False
    ref = ops.resblock_chain(blocks, x, out=mkout())
    torch.cuda.synchronize()
    ops.CONV_CHAIN = True
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    bad = 0
    for r in range(reps):
        with ops.range_status(st):
            got = ops.resblock_chain(blocks, x, out=mkout())
        torch.cuda.synchronize()
        if not torch.equal(got, ref):
            bad += 1
            d = (got - ref).abs()
            print("   MISMATCH rep", r, "max", float(d.max()), "n wrong", int((d > 0).sum()), "nan", int(torch.isnan(got).sum()))
    def tm(flag):
        ops.CONV_CHAIN = flag
        for _ in range(2): ops.resblock_chain(blocks, x)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.resblock_chain(blocks, x)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 5
    t0, t1 = tm(False), tm(True)
    print("N %d C %d %dx%d blocks %2d strided %d: %s  status %d  per-layer %.3f ms  chain %.3f ms" % (n, c, h, w, nb, strided, "bits equal" if not bad else "WRONG", int(st.item()), t0, t1), flush=True)
    return bad
bad = 0
bad += run(1, 64, 16, 32, 1)
bad += run(1, 64, 40, 64, 2)
bad += run(2, 64, 36, 100, 3)
bad += run(3, 64, 180, 320, 2)
bad += run(2, 64, 180, 320, 5, strided=True)
bad += run(3, 64, 180, 320, 40, reps=5)
bad += run(1, 64, 45, 80, 40, reps=5)
bad += run(1, 56, 64, 64, 4)
print("FAILED" if bad else "all equal")
