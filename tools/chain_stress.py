"""Chain launches (ops.resblock_chain) of four shapes on three HIP streams at once, beside other work: every result against the single launches, bit for bit.
   python tools/chain_stress.py"""
import sys, torch
sys.path.insert(0, ".")
from motif_amd import ops
torch.manual_seed(0)
dev = "cuda"
def mk(c, nb):
    return [tuple(ops.ConvPlan(torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5), torch.randn(c, device=dev) * 0.1, 1, 1, 1, 1, 0) for _ in range(2)) for _ in range(nb)]
cases = [(3, 180, 320, 40), (2, 180, 320, 5), (6, 180, 320, 12), (3, 256, 448, 8)]
blocks = [mk(64, nb) for (_, _, _, nb) in cases]
xs = [torch.randn(n, 64, h, w, device=dev) for (n, h, w, _) in cases]
ops.CONV_CHAIN = False
refs = [ops.resblock_chain(b, x) for b, x in zip(blocks, xs)]
ops.CONV_CHAIN = True
[ops.resblock_chain(b, x) for b, x in zip(blocks, xs)]
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(3)]
st = torch.zeros(1, dtype=torch.int32, device=dev)
noise = torch.randn(64, 1 << 20, device=dev)
bad = 0
for it in range(150):
    outs = []
    for k, s_ in enumerate(streams):
        i = (it + k) % len(cases)
        with torch.cuda.stream(s_), ops.range_status(st):
            if k == 2:
                (noise * 1.0001).sum()                       # other work competing for the CUs
            outs.append((i, ops.resblock_chain(blocks[i], xs[i])))
    torch.cuda.synchronize()
    for i, o in outs:
        if not torch.equal(o, refs[i]):
            bad += 1
print("stress: %d mismatching results of %d chain launches on three streams; status %d" % (bad, 150 * 3, int(st.item())))
