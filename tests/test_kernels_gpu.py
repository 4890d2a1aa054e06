"""GPU parity tests, one per C-ABI kernel family: HIP path vs the CPU oracle / torch CPU on seeded inputs.

Bit-exact where the domain is integer/index (splat count & max, corner indices), fp32 tolerance stated
per test elsewhere.  All calls go through the C ABI (motif_amd.ops -> libmotif_hip.so).
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def close(a, b, atol, rtol=0.0, what=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bad.any(), "%s: max|diff|=%.3e (tol %.1e) at %d/%d elements, ref max %.3e" % (
        what, err.max().item(), atol, int(bad.sum()), bad.numel(), b.abs().max().item())


@pytest.fixture(params=["f16x2", "bf16x3", "fp32"])
def engine(request):
    """Arithmetic engine of the dense contractions for one test: the 3-way bf16 split on the bf16 matrix cores
    (conv_split_kernel / dcn_fused_kernel<8,true>), the same with the two-part fp16 form of conv_wino.hip on the 3x3 stride-1
    layers ("f16x2", the bench default) and the fp32 MFMA (conv_igemm_kernel / dcn_fused_kernel<8,false>).  The previous mode is restored afterwards -- no test leaks its mode into the next."""
    from motif_amd import ops
    before = ops.get_mma()
    ops.set_mma(request.param)
    try:
        yield request.param
    finally:
        ops.set_mma(before)


@pytest.fixture
def keep_mma():
    """For tests that switch engines themselves: put back whatever was selected before."""
    from motif_amd import ops
    conv, siren = ops.get_conv_mma(), ops.get_siren_mma()
    try:
        yield
    finally:
        ops.set_conv_mma(conv)
        ops.set_siren_mma(siren)


# ------------------------------------------------------------------------------------------- conv engine
CONV_CASES = [
    # cin, cout, k, stride, pad, dil, groups, pad_mode, H, W, N
    (64, 64, 3, 1, 1, 1, 1, "zeros", 45, 80, 1),
    (64, 64, 3, 2, 1, 1, 1, "zeros", 46, 84, 2),
    (3, 64, 3, 1, 1, 1, 1, "zeros", 33, 47, 1),
    (128, 64, 1, 1, 0, 1, 1, "zeros", 20, 40, 1),
    (3, 32, 7, 2, 3, 1, 1, "zeros", 64, 96, 2),
    (14, 64, 3, 1, 1, 1, 2, "zeros", 32, 32, 2),
    (64, 64, 3, 1, 1, 1, 2, "zeros", 19, 33, 1),
    (64, 64, 3, 1, 1, 1, 1, "reflect", 32, 40, 1),
    (64, 216, 3, 1, 1, 1, 1, "zeros", 24, 32, 1),
    (128, 256, 3, 1, 1, 1, 1, "zeros", 16, 32, 1),
    (96, 128, 3, 1, 1, 1, 1, "zeros", 16, 20, 1),
    (242, 96, 3, 1, 1, 1, 1, "zeros", 16, 24, 1),
    (196, 96, 1, 1, 0, 1, 1, "zeros", 16, 24, 1),
    (2, 64, 7, 1, 3, 1, 1, "zeros", 16, 24, 1),
    (128, 2, 3, 1, 1, 1, 1, "zeros", 16, 24, 1),
    (32, 8, 1, 1, 0, 1, 1, "zeros", 40, 64, 1),
    (8, 8, 3, 2, 1, 1, 1, "zeros", 40, 64, 1),
    (32, 64, 1, 2, 0, 1, 1, "zeros", 40, 64, 1),
    (128, 128, 3, 1, 2, 2, 1, "zeros", 24, 32, 1),
    (96, 64, 3, 1, 16, 16, 1, "zeros", 24, 40, 1),
    (565, 128, 3, 1, 1, 1, 1, "zeros", 12, 20, 1),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv2d_matches_torch_cpu(case, engine):
    from motif_amd.models.modules.layers import Conv2d
    cin, cout, k, stride, pad, dil, groups, pm, H, W, N = case
    m = Conv2d(cin, cout, k, stride, pad, dil, groups, True, pm)
    with torch.no_grad():
        m.weight.copy_(rnd(*m.weight.shape, seed=1, scale=1.0 / math.sqrt(cin * k * k / groups)))
        m.bias.copy_(rnd(cout, seed=2, scale=0.1))
    x = rnd(N, cin, H, W, seed=3)
    xp = F.pad(x, (pad,) * 4, mode="reflect") if pm == "reflect" else x
    ref = F.conv2d(xp, m.weight, m.bias, stride, 0 if pm == "reflect" else pad, dil, groups)
    m = m.to(dev())
    out = m(x.to(dev()))
    close(out, ref, 2e-5, 2e-5, "conv")


@pytest.mark.parametrize("case", [(32, 8, 1, 1, 0, 1, 40, 64, 2), (3, 32, 7, 2, 3, 1, 64, 96, 2), (8, 8, 3, 2, 1, 1, 40, 64, 1),
                                  (128, 2, 3, 1, 1, 1, 16, 24, 1), (96, 64, 3, 1, 16, 16, 24, 40, 1), (14, 24, 3, 1, 1, 1, 30, 52, 2)])
def test_conv_igemm_16_byte_staging_is_bit_identical_to_the_scalar_plan(case, keep_mma):
    """The fp32 engine's VEC staging (4 pixels per lane, LDS rows shifted to aligned quads) changes how the patch reaches LDS,
    not a single product: outputs must equal the 4-byte plan (option conv_novec) bit for bit -- 1x1, 7x7 stride 2, 3x3 stride 2,
    a 2-cout head, a dilated layer, odd channel count; zero padding of every width."""
    from motif_amd import ops
    from motif_amd.models.modules.layers import Conv2d
    cin, cout, k, stride, pad, dil, H, W, N = case
    ops.set_conv_mma(ops.MMA_FP32)
    m = Conv2d(cin, cout, k, stride, pad, dil, 1, True, "zeros")
    with torch.no_grad():
        m.weight.copy_(rnd(*m.weight.shape, seed=1, scale=1.0 / math.sqrt(cin * k * k)))
        m.bias.copy_(rnd(cout, seed=2, scale=0.1))
    x = rnd(N, cin, H, W, seed=3).to(dev())
    m = m.to(dev())
    try:
        a = m(x, act=ops.ACT_RELU).clone()
        ops.set_option("conv_novec", 1)
        b = m(x, act=ops.ACT_RELU).clone()
    finally:
        ops.set_option("conv_novec", 0)
    assert torch.equal(a, b)
    ref = F.relu(F.conv2d(x.cpu(), m.weight.cpu(), m.bias.cpu(), stride, pad, dil))
    close(a, ref, 2e-5, 2e-5, "vec staging vs torch")


def test_conv2d_fused_epilogues_and_concat(engine):
    from motif_amd import ops
    from motif_amd.models.modules.layers import Conv2d
    m = Conv2d(96 + 40, 80, 3, 1, 1)
    with torch.no_grad():
        m.weight.copy_(rnd(*m.weight.shape, seed=1, scale=0.05))
        m.bias.copy_(rnd(80, seed=2, scale=0.1))
    a, b = rnd(2, 96, 20, 36, seed=3), rnd(2, 40, 20, 36, seed=4)
    res = rnd(2, 80, 20, 36, seed=5)
    base = F.conv2d(torch.cat([a, b], 1), m.weight, m.bias, 1, 1)
    m = m.to(dev())
    ad, bd, rd = a.to(dev()), b.to(dev()), res.to(dev())
    close(m(ad, bd, act=ops.ACT_LRELU), F.leaky_relu(base, 0.1), 2e-5, 2e-5, "concat+lrelu")
    close(m(ad, bd, act=ops.ACT_SIGMOID), torch.sigmoid(base), 5e-6, 0, "sigmoid")
    close(m(ad, bd, act=ops.ACT_TANH), torch.tanh(base), 5e-6, 0, "tanh")
    close(m(ad, bd, res=rd, res_mode=1), base + res, 2e-5, 2e-5, "res add")
    close(m(ad, bd, act=ops.ACT_RELU, res=rd, res_mode=3), F.relu(F.relu(base) + res), 2e-5, 2e-5, "relu(relu+res)")
    close(m(ad, bd, act=ops.ACT_SIGMOID, res=rd, res_mode=4), torch.sigmoid(base) * res, 2e-6, 0, "sigmoid*res")
    split = torch.cat([torch.tanh(base[:, :48]), F.relu(base[:, 48:])], 1)
    close(m(ad, bd, act=ops.ACT_TANH, act2=ops.ACT_RELU, act_split=48), split, 2e-5, 2e-5, "act split")
    # strided batch views in and out
    big = torch.zeros(2, 3, 96, 20, 36, device=dev())
    big[:, 1] = ad
    outbuf = torch.zeros(2, 100, 20, 36, device=dev())
    m(big[:, 1], bd, out=outbuf[:, 10:90])
    close(outbuf[:, 10:90], base, 2e-5, 2e-5, "strided views")
    assert float(outbuf[:, :10].abs().max()) == 0 and float(outbuf[:, 90:].abs().max()) == 0


SPLIT_CASES = [  # cin, cout, groups, pad_mode, H, W, N, two-source split (0 = single input)
    (64, 64, 1, "zeros", 45, 80, 2, 0),
    (128, 64, 1, "zeros", 23, 37, 1, 64),     # concat of two sources, ragged tile edges
    (64, 216, 1, "zeros", 20, 36, 1, 0),      # partial cout group (216 = 3*64 + 24)
    (48, 96, 1, "reflect", 19, 33, 1, 0),     # 3 channel chunks, reflect padding
    (40, 80, 2, "zeros", 17, 40, 2, 0),       # groups: 20 -> 32 padded channels per group, 40 couts per group
]


@pytest.mark.parametrize("case", SPLIT_CASES)
def test_conv_split_engine_is_fp32_equivalent(case, keep_mma):
    """mma=6 (3-way bf16 split, 6 products on the bf16 matrix cores) and mma=7 (2-way fp16 split, 3 products) against an fp64
    convolution: their error must not exceed the fp32-MFMA engine's; mma=3 / mma=1 keep 16 / 8 mantissa bits."""
    from motif_amd import ops
    from motif_amd.models.modules.layers import Conv2d
    cin, cout, groups, pm, H, W, N, c0 = case
    m = Conv2d(cin, cout, 3, 1, 1, 1, groups, True, pm)
    with torch.no_grad():
        m.weight.copy_(rnd(*m.weight.shape, seed=1, scale=1.0 / math.sqrt(cin * 9 / groups)))
        m.bias.copy_(rnd(cout, seed=2, scale=0.1))
    x = rnd(N, cin, H, W, seed=3)
    res = rnd(N, cout, H, W, seed=4)
    xp = F.pad(x.double(), (1,) * 4, mode="reflect") if pm == "reflect" else x.double()
    ref = F.leaky_relu(F.conv2d(xp, m.weight.double(), m.bias.double(), 1, 0 if pm == "reflect" else 1, 1, groups) + res.double(), 0.1)
    m = m.to(dev())
    xd, rd = x.to(dev()), res.to(dev())
    args = (xd[:, :c0].contiguous(), xd[:, c0:].contiguous()) if c0 else (xd, None)
    err = {}
    for mode in (ops.MMA_FP32, ops.MMA_BF16X3, ops.MMA_F16X2, ops.MMA_BF16X2, ops.MMA_BF16):
        ops.set_conv_mma(mode)
        out = m(*args, act=ops.ACT_LRELU, res=rd, res_mode=1)
        err[mode] = float((out.detach().double().cpu() - ref).abs().max())
    scale = float(ref.abs().max())
    assert err[ops.MMA_FP32] < 2e-6 * scale, err
    assert err[ops.MMA_BF16X3] <= 1.25 * err[ops.MMA_FP32] + 1e-7 * scale, err
    # two fp16 parts, three products (conv_wino.hip where the layer is eligible, the three-part kernels elsewhere): the same bound
    assert err[ops.MMA_F16X2] <= 1.25 * err[ops.MMA_FP32] + 1e-7 * scale, err
    assert err[ops.MMA_BF16X2] < 1e-4 * scale, err
    assert 1e-4 * scale < err[ops.MMA_BF16] < 3e-2 * scale, err      # really ran in bf16


DIRECT_CASES = [  # cin, cout, k, H, W, N, two-source split, act, res_mode, act_split
    (32, 8, 1, 360, 640, 2, 0, "relu", 0, 0),         # RAFT-small bottleneck conv1 at half resolution
    (8, 8, 3, 360, 640, 2, 0, "relu", 0, 0),          # conv2 (3x3): image borders, halo columns
    (8, 16, 1, 360, 640, 2, 0, "none", 2, 0),         # two cout slices of 8, residual after
    (32, 16, 1, 200, 332, 2, 16, "lrelu", 1, 0),      # two-source input, residual before the activation
    (5, 12, 3, 190, 364, 2, 0, "tanh", 3, 8),         # odd channel count (zero partner channel in the packed block), partial cout slice, split activation
    (16, 24, 3, 96, 96, 15, 0, "sigmoid", 4, 0),      # many small images, multiplicative residual (24 couts: stays on the MFMA engine)
    (128, 2, 3, 90, 160, 2, 0, "none", 0, 0),         # RAFT flow head: deep-K form (8 channel slices per workgroup, LDS reduction)
    (200, 3, 3, 46, 76, 2, 104, "lrelu", 2, 0),       # deep form: two sources, slice ends inside a source, ragged quads, 3 couts
    (96, 4, 1, 128, 132, 1, 0, "relu", 1, 0),         # deep form, 1x1
    (529, 2, 3, 12, 20, 1, 0, "none", 0, 0),          # PWC-Net flow head on the coarsest level: 16-slice deep form (one workgroup)
    (597, 2, 3, 48, 80, 2, 0, "none", 0, 0),          # the same form, several workgroups, ragged channel slices
    (300, 1, 3, 20, 24, 1, 0, "relu", 0, 0),          # one cout
]


@pytest.mark.parametrize("case", DIRECT_CASES)
def test_conv_direct_narrow_layers_match_the_mfma_engine_and_torch(case):
    """conv_direct.hip (narrow layers on large maps: four pixels x <= 16 couts per thread, scalar-loaded weights from the block
    conv_igemm packs) against an fp64 reference and against the MFMA engine on the same launch (option conv_nodirect)."""
    from motif_amd import ops
    from motif_amd.models.modules.layers import Conv2d
    cin, cout, k, H, W, N, c0, actn, rm, asplit = case
    acts = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU, "tanh": ops.ACT_TANH, "sigmoid": ops.ACT_SIGMOID}
    fact = {"none": lambda v: v, "relu": F.relu, "lrelu": lambda v: F.leaky_relu(v, 0.1), "tanh": torch.tanh, "sigmoid": torch.sigmoid}[actn]
    m = Conv2d(cin, cout, k, 1, k // 2)
    with torch.no_grad():
        m.weight.copy_(rnd(*m.weight.shape, seed=1, scale=1.0 / math.sqrt(cin * k * k)))
        m.bias.copy_(rnd(cout, seed=2, scale=0.1))
    x, res = rnd(N, cin, H, W, seed=3), rnd(N, cout, H, W, seed=4)
    y = F.conv2d(x.double(), m.weight.double(), m.bias.double(), 1, k // 2)
    r = res.double()
    if rm == 1: y = y + r
    y = torch.cat([fact(y[:, :asplit]), F.relu(y[:, asplit:])], 1) if asplit else fact(y)
    if rm == 2: y = y + r
    elif rm == 3: y = F.relu(y + r)
    elif rm == 4: y = y * r
    m = m.to(dev())
    xd, rd = x.to(dev()), res.to(dev())
    args = (xd[:, :c0].contiguous(), xd[:, c0:].contiguous()) if c0 else (xd, None)
    kw = dict(act=acts[actn], res=rd if rm else None, res_mode=rm)
    if asplit: kw.update(act2=ops.ACT_RELU, act_split=asplit)
    scale = float(y.abs().max())
    try:
        ops.set_option("conv_nodirect", 1)
        mf = m(*args, **kw).double().cpu()
        ops.set_option("conv_nodirect", 0)
        di = m(*args, **kw).double().cpu()
    finally:
        ops.set_option("conv_nodirect", 0)
    tol = 2e-6 * scale * max(1.0, math.sqrt(cin * k * k / 1800.0))      # fp32 accumulation: the bound grows with the square root of the reduction length
    assert float((mf - y).abs().max()) < tol
    assert float((di - y).abs().max()) < tol
    assert float((di - mf).abs().max()) < tol


def test_conv_split_multi_problem_and_views(keep_mma):
    from motif_amd import ops
    from motif_amd.models.modules.layers import Conv2d
    ms = [Conv2d(64, 64, 3, 1, 1) for _ in range(3)]
    xs = [rnd(2, 64, 21, 50, seed=10 + i) for i in range(3)]
    with torch.no_grad():
        for i, m in enumerate(ms):
            m.weight.copy_(rnd(64, 64, 3, 3, seed=20 + i, scale=0.04))
            m.bias.copy_(rnd(64, seed=30 + i, scale=0.1))
    refs = [F.relu(F.conv2d(x, m.weight, m.bias, 1, 1)) for m, x in zip(ms, xs)]
    ms = [m.to(dev()) for m in ms]
    ops.set_conv_mma(ops.MMA_BF16X3)
    out = ops.conv2d_multi([m.plan() for m in ms], [x.to(dev()) for x in xs], act=ops.ACT_RELU)
    for i in range(3):
        close(out[i], refs[i], 2e-5, 2e-5, "split multi %d" % i)
    big = torch.zeros(2, 2, 64, 21, 50, device=dev())
    big[:, 1] = xs[0].to(dev())
    buf = torch.zeros(2, 80, 21, 50, device=dev())
    ms[0](big[:, 1], out=buf[:, 8:72], act=ops.ACT_RELU)
    close(buf[:, 8:72], refs[0], 2e-5, 2e-5, "split strided views")
    assert float(buf[:, :8].abs().max()) == 0 and float(buf[:, 72:].abs().max()) == 0


@pytest.mark.parametrize("tile", [2, 3, 4, 5, 7], ids=["rows12", "rows8", "rows6x2", "wino", "wino_f16x2"])
@pytest.mark.parametrize("shape", [(3, 64, 64, 180, 320), (2, 64, 216, 90, 160), (5, 128, 64, 63, 100), (1, 48, 80, 19, 36), (2, 81, 96, 12, 16)])
def test_conv_split2_persistent_tiles(shape, tile, keep_mma):
    """The round-3 conv kernel over MANY tiles per workgroup (persistent loop, next tile staged under the last chunk, ragged last
    tile row / column, partial cout group, channel padding) against torch on the host and against the round-2 kernel.  The 81-channel
    case (PWC-Net's first decoder layer) has a ragged last 16-channel chunk at the very end of the tensor: the staging loads are
    range-checked buffer loads, so the 15 missing planes read as zeros instead of reading (and once faulting) past the allocation."""
    from motif_amd import ops
    from motif_amd.models.modules.layers import Conv2d
    n, cin, cout, H, W = shape
    m = Conv2d(cin, cout, 3, 1, 1)
    with torch.no_grad():
        m.weight.copy_(rnd(*m.weight.shape, seed=1, scale=1.0 / math.sqrt(cin * 9)))
        m.bias.copy_(rnd(cout, seed=2, scale=0.1))
    x, res = rnd(n, cin, H, W, seed=3), rnd(n, cout, H, W, seed=4)
    ref = F.relu(F.conv2d(x, m.weight, m.bias, 1, 1)) + res
    m = m.to(dev())
    ops.set_conv_mma(ops.MMA_F16X2 if tile == 7 else ops.MMA_BF16X3)
    try:
        ops.set_option("conv_engine", 5 if tile == 7 else tile)      # the round-3 kernel, 12-row (2) or 8-row (3) tiles, whatever the tile count; 5 = the round-4 Winograd F(2,3) kernel (7: its two-part fp16 form)
        out = m(x.to(dev()), act=ops.ACT_RELU, res=res.to(dev()), res_mode=2)
        ops.set_conv_mma(ops.MMA_BF16X3)
        ops.set_option("conv_engine", 1)
        old = m(x.to(dev()), act=ops.ACT_RELU, res=res.to(dev()), res_mode=2)
    finally:
        ops.set_option("conv_engine", 0)
    close(out, ref, 2e-5, 2e-5, "split2 vs torch")
    tol = 4e-6 if tile in (5, 7) else 2e-6                    # Winograd F(2,3): one more fp32 addition per operand and a three-term output sum
    close(out, old, tol, tol, "split2 / wino vs two-block kernel")


@pytest.mark.parametrize("case", [(6, 64, 64, 180, 320, "lrelu", 0), (3, 64, 64, 64, 96, "relu", 2), (3, 128, 64, 64, 96, "relu", 0), (3, 64, 216, 64, 96, "none", 0),
                                  (5, 64, 64, 16, 24, "none", 1)], ids=lambda c: "x".join(str(v) for v in c))
@pytest.mark.parametrize("parts", [3, 2], ids=["bf16x3", "f16x2"])
def test_conv_wino_same_bits_whatever_the_batch_and_the_run(case, parts, keep_mma):
    """A frame's output bits depend neither on the batch it sits in (which workgroup picks a tile up, what that workgroup did before)
    nor on the run.  Found a timing-dependent fault the tolerance tests let through: gfx950 wants TWO wait states between a store of
    more than 8 bytes and the next vector write of its data registers, and hipcc adds none around inline-assembly stores."""
    from motif_amd import ops
    from motif_amd.models.modules.layers import Conv2d
    n, cin, cout, H, W, actn, rm = case
    act = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU}[actn]
    m = Conv2d(cin, cout, 3, 1, 1)
    with torch.no_grad():
        m.weight.copy_(rnd(*m.weight.shape, seed=1, scale=1.0 / math.sqrt(cin * 9)))
        m.bias.copy_(rnd(cout, seed=2, scale=0.1))
    m = m.to(dev())
    x, res = rnd(n, cin, H, W, seed=3).to(dev()), rnd(n, cout, H, W, seed=4).to(dev())
    ops.set_conv_mma(ops.MMA_BF16X3 if parts == 3 else ops.MMA_F16X2)
    try:
        ops.set_option("conv_engine", 5)
        kw = lambda k: dict(act=act) if rm == 0 else dict(act=act, res=res[:k].contiguous(), res_mode=rm)
        full = m(x, **kw(n)).clone()
        for rep in range(8):
            assert torch.equal(m(x, **kw(n)), full), "run %d differs from run 0" % rep
        for k in range(1, n):
            assert torch.equal(m(x[:k].contiguous(), **kw(k)), full[:k]), "the first %d frames alone differ from the same frames inside the batch of %d" % (k, n)
    finally:
        ops.set_option("conv_engine", 0)


PW_CASES = [  # N, Cin, Cout, H, W, first-source channels (0 = one source), act, res_mode
    (2, 128, 64, 45, 80, 0, "lrelu", 0),          # the fusion layers' shape class
    (1, 196, 96, 23, 37, 0, "none", 1),           # PWC / RAFT feature head: three cout tiles, ragged last pixel group, Cin % 16 != 0
    (2, 24, 96, 30, 40, 0, "relu", 2),            # two k-steps, the second half empty past channel 24
    (3, 128, 64, 16, 24, 64, "lrelu", 0),         # two concatenated sources
    (1, 64, 128, 18, 32, 0, "none", 0),           # four cout tiles
    (2, 16, 40, 33, 31, 0, "relu", 1),            # one k-step, partial second cout tile, odd plane size
]


@pytest.mark.parametrize("case", PW_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv_pw_pointwise_layers_match_fp64_and_the_fp32_engine(case, keep_mma):
    """conv_pw.hip (1x1 layers under mma = 7: two fp16 parts on the matrix cores, weights split in the kernel from the fp32 packed
    block) against an fp64 convolution -- error not above the fp32-MFMA engine's, as for the other split kernels -- and against that engine."""
    from motif_amd import ops
    from motif_amd.models.modules.layers import Conv2d
    n, cin, cout, H, W, c0, actn, rm = case
    act = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU}[actn]
    m = Conv2d(cin, cout, 1, 1, 0)
    with torch.no_grad():
        m.weight.copy_(rnd(*m.weight.shape, seed=1, scale=1.0 / math.sqrt(cin)))
        m.bias.copy_(rnd(cout, seed=2, scale=0.1))
    x, res = rnd(n, cin, H, W, seed=3), rnd(n, cout, H, W, seed=4)
    y = F.conv2d(x.double(), m.weight.double(), m.bias.double())
    f = {"none": lambda v: v, "relu": F.relu, "lrelu": lambda v: F.leaky_relu(v, 0.1)}[actn]
    ref = f(y + res.double()) if rm == 1 else f(y) + res.double() if rm == 2 else f(y)
    m = m.to(dev())
    xd, rd = x.to(dev()), res.to(dev())
    args = (xd[:, :c0].contiguous(), xd[:, c0:].contiguous()) if c0 else (xd, None)
    kw = dict(act=act) if rm == 0 else dict(act=act, res=rd, res_mode=rm)
    outs = {}
    for mode in (ops.MMA_FP32, ops.MMA_F16X2):
        ops.set_conv_mma(mode)
        outs[mode] = m(*args, **kw).double().cpu()
    e32, e16 = float((outs[ops.MMA_FP32] - ref).abs().max()), float((outs[ops.MMA_F16X2] - ref).abs().max())
    scale = float(ref.abs().max())
    assert not torch.equal(outs[ops.MMA_FP32], outs[ops.MMA_F16X2]), "the two engines are different kernels"
    assert e32 < 2e-6 * scale and e16 <= 1.25 * e32 + 1e-7 * scale, (e32, e16, scale)
    # same bits whatever the batch: a pixel group's arithmetic does not depend on which wave picks it up
    again = m(*(t[:1].contiguous() if t is not None else None for t in args), **(dict(act=act) if rm == 0 else dict(act=act, res=rd[:1].contiguous(), res_mode=rm)))
    assert torch.equal(again.double().cpu(), outs[ops.MMA_F16X2][:1])


@pytest.mark.parametrize("case", [(3, 64, 64, 45, 80, "relu", 1), (2, 128, 64, 37, 52, "lrelu", 0), (1, 64, 216, 20, 36, "none", 0), (2, 242, 96, 23, 40, "tanh", 4),
                                  (2, 48, 40, 19, 36, "none", 2)])
def test_conv_wino_transposed_accumulators_give_the_bits_of_the_row_major_form(case, keep_mma):
    """Round 5: conv_wino_kernel<2, ., TR = true> swaps the two MFMA operands, so its accumulators come out transposed (lane = cout,
    registers = pixels) and the epilogue stores 16-byte row pieces straight from registers.  Same products, same fp32 sums in the
    same order, same epilogue arithmetic: the output must equal the row-major form BIT FOR BIT -- full and ragged tiles, partial cout
    groups, every residual mode, a transcendental activation.  (Opt-in, option conv_wino_tr = 1: measured no faster)"""
    from motif_amd import ops
    from motif_amd.models.modules.layers import Conv2d
    n, cin, cout, H, W, actn, rm = case
    act = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU, "tanh": ops.ACT_TANH}[actn]
    m = Conv2d(cin, cout, 3, 1, 1)
    with torch.no_grad():
        m.weight.copy_(rnd(*m.weight.shape, seed=1, scale=1.0 / math.sqrt(cin * 9)))
        m.bias.copy_(rnd(cout, seed=2, scale=0.1))
    m = m.to(dev())
    x, res = rnd(n, cin, H, W, seed=3).to(dev()), rnd(n, cout, H, W, seed=4).to(dev())
    kw = dict(act=act) if rm == 0 else dict(act=act, res=res, res_mode=rm)
    ops.set_conv_mma(ops.MMA_F16X2)
    outs = {}
    try:
        ops.set_option("conv_engine", 5)
        for tr in (1, 0):
            ops.set_option("conv_wino_tr", tr)
            outs[tr] = m(x, **kw).clone()
    finally:
        ops.set_option("conv_wino_tr", 0)
        ops.set_option("conv_engine", 0)
    assert torch.equal(outs[0], outs[1])
    ref = F.conv2d(x.double().cpu(), m.weight.double().cpu(), m.bias.double().cpu(), 1, 1)
    assert float((outs[0].double().cpu() - (torch.tanh(ref) * res.double().cpu() if rm == 4 else ref)).abs().max()) < 1e3      # (finite; the value tests are the engine tests above)


def _chain_blocks(c, nb, seed):
    from motif_amd import ops
    blocks = []
    for b in range(nb):
        pl = []
        for j in range(2):
            w = rnd(c, c, 3, 3, seed=seed + 10 * b + j, scale=1.0 / (3.0 * math.sqrt(c))).to(dev())
            pl.append(ops.ConvPlan(w, rnd(c, seed=seed + 10 * b + j + 5, scale=0.1).to(dev()), 1, 1, 1, 1, 0))
        blocks.append(tuple(pl))
    return blocks


@pytest.mark.parametrize("case", [(1, 64, 16, 32, 1, False), (1, 64, 40, 64, 2, False), (2, 64, 36, 100, 3, False), (3, 64, 180, 320, 4, False), (2, 64, 180, 320, 5, True),
                                  (1, 56, 61, 64, 4, False), (1, 64, 45, 80, 12, False), (3, 64, 90, 160, 40, False)])
def test_conv_chain_gives_the_bits_of_the_launches_layer_by_layer(case, keep_mma):
    """Round 5, VERDICT r4 "trunk fusion": `motif_conv2d_chain_fwd` runs the 2 B convolutions of B residual blocks (module_util.py:34-52 in an
    nn.Sequential, Ours.py:349-356) as ONE persistent launch -- tiles of all layers in ticket order, a tile waiting for the row words of the
    layer before.  The per-tile computation is conv_wino's, so the result must equal B x 2 calls of motif_conv2d_fwd BIT FOR BIT: one tile,
    fewer tiles than CUs (every tile waits for its producers), ragged tiles, C = 56, a batch-strided output view, the trunk's 80 layers; three
    times over (the dependency order differs from run to run), with a clean status word."""
    from motif_amd import ops
    n, c, H, W, nb, strided = case
    ops.set_conv_mma(ops.MMA_F16X2)
    blocks = _chain_blocks(c, nb, 7)
    x = rnd(n, c, H, W, seed=99).to(dev())

    def out_view():
        return torch.zeros(n, 2, c, H, W, device=dev())[:, 0] if strided else None
    saved = ops.CONV_CHAIN, ops.CONV_CHAIN_MIN_TILES
    st = torch.zeros(1, dtype=torch.int32, device=dev())
    try:
        ops.CONV_CHAIN = False
        ref = ops.resblock_chain(blocks, x, out=out_view())
        ops.CONV_CHAIN, ops.CONV_CHAIN_MIN_TILES = True, 1
        calls = []
        orig = ops.conv2d_chain
        ops.conv2d_chain = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            for _ in range(3):
                with ops.range_status(st):
                    got = ops.resblock_chain(blocks, x, out=out_view())
                assert torch.equal(got, ref)
        finally:
            ops.conv2d_chain = orig
        assert len(calls) == 3, "the chain entry did not take the shape"
        assert int(st.item()) == 0
    finally:
        ops.CONV_CHAIN, ops.CONV_CHAIN_MIN_TILES = saved
    y = x.double().cpu()
    for p1, p2 in blocks:       # (value check against fp64: the engine tests above hold the arithmetic; this guards the buffer rotation)
        t = F.relu(F.conv2d(y, p1.weight.double().cpu(), p1.bias.double().cpu(), 1, 1))
        y = y + F.conv2d(t, p2.weight.double().cpu(), p2.bias.double().cpu(), 1, 1)
    assert float((ref.double().cpu() - y).abs().max()) <= 2e-5 * max(1.0, float(y.abs().max())) * math.sqrt(nb)


def test_conv_chain_lateral_blocks_leaky_relu_and_an_activation_on_the_last_sum(keep_mma):
    """The five LateralBlocks of `flow_process` (Ours.py:425-431): conv - leaky ReLU - conv + x, the LAST block's sum through a leaky ReLU
    (epilogue = activation of (value + residual)): one chain launch, the bits of the ten launches."""
    from motif_amd import ops
    ops.set_conv_mma(ops.MMA_F16X2)
    blocks = _chain_blocks(64, 5, 40)
    x = rnd(2, 64, 90, 160, seed=17).to(dev())
    saved = ops.CONV_CHAIN, ops.CONV_CHAIN_MIN_TILES
    try:
        ops.CONV_CHAIN = False
        ref = ops.resblock_chain(blocks, x, act=ops.ACT_LRELU, last_act=ops.ACT_LRELU)
        ops.CONV_CHAIN, ops.CONV_CHAIN_MIN_TILES = True, 1
        got = ops.resblock_chain(blocks, x, act=ops.ACT_LRELU, last_act=ops.ACT_LRELU)
    finally:
        ops.CONV_CHAIN, ops.CONV_CHAIN_MIN_TILES = saved
    assert torch.equal(got, ref)
    y = x.double().cpu()
    for i, (p1, p2) in enumerate(blocks):
        t = F.leaky_relu(F.conv2d(y, p1.weight.double().cpu(), p1.bias.double().cpu(), 1, 1), 0.1)
        y = y + F.conv2d(t, p2.weight.double().cpu(), p2.bias.double().cpu(), 1, 1)
        if i == 4:
            y = F.leaky_relu(y, 0.1)
    assert float((ref.double().cpu() - y).abs().max()) <= 5e-5 * max(1.0, float(y.abs().max()))


def test_conv_chain_two_chains_in_flight_and_an_out_of_range_operand(keep_mma):
    """Two chain launches on two streams at once (the clips in flight of bench.py): a workgroup that is not resident holds no ticket, so
    neither launch can starve the other -- both finish and both are exact.  Then ONE feature of 1e5 in the input: the chain's kernels report
    it in the status word (bit 0) like the single launches do."""
    from motif_amd import ops
    ops.set_conv_mma(ops.MMA_F16X2)
    blocks = [_chain_blocks(64, 6, 300), _chain_blocks(64, 6, 500)]
    xs = [rnd(3, 64, 96, 160, seed=31).to(dev()), rnd(3, 64, 96, 160, seed=32).to(dev())]
    saved = ops.CONV_CHAIN, ops.CONV_CHAIN_MIN_TILES
    try:
        ops.CONV_CHAIN = False
        refs = [ops.resblock_chain(b, x) for b, x in zip(blocks, xs)]
        ops.CONV_CHAIN, ops.CONV_CHAIN_MIN_TILES = True, 1
        [ops.resblock_chain(b, x) for b, x in zip(blocks, xs)]           # tables built, weights packed
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        for rep in range(4):
            outs = []
            for s_, b, x in zip(streams, blocks, xs):
                with torch.cuda.stream(s_):
                    outs.append(ops.resblock_chain(b, x))
            torch.cuda.synchronize()
            assert torch.equal(outs[0], refs[0]) and torch.equal(outs[1], refs[1])
        st = torch.zeros(1, dtype=torch.int32, device=dev())
        xb = xs[0].clone()
        xb[1, 5, 40, 77] = 1.0e5
        with ops.range_status(st):
            ops.resblock_chain(blocks[0], xb)
        assert int(st.item()) & 1
    finally:
        ops.CONV_CHAIN, ops.CONV_CHAIN_MIN_TILES = saved


def test_conv_chain_abandoned_or_mis_tabled_launch_reports_and_touches_nothing(keep_mma):
    """(i) a chain that gives up (here forced: option conv_dbg bit 7 makes every workgroup give up at its first tile -- in operation
    that takes a second without ANY tile of the chain being published) must be observable: status bit 1, also through the fallback word a
    launch outside `ops.range_status` gets (`ops.check_chain_status` raises); (ii) a layer table naming a scratch buffer the workspace does not
    hold (`work_floats` too small for the documented three-buffer rotation) is refused ON THE DEVICE before anything is read or written:
    status bit 2, the output untouched; a valid call afterwards is exact."""
    import ctypes
    from motif_amd import ops, _lib
    ops.set_conv_mma(ops.MMA_F16X2)
    blocks = _chain_blocks(64, 3, 70)
    x = rnd(2, 64, 40, 64, seed=5).to(dev())
    saved = ops.CONV_CHAIN, ops.CONV_CHAIN_MIN_TILES
    try:
        ops.CONV_CHAIN = False
        ref = ops.resblock_chain(blocks, x)
        ops.CONV_CHAIN, ops.CONV_CHAIN_MIN_TILES = True, 1
        st = torch.zeros(1, dtype=torch.int32, device=dev())
        try:
            ops.set_option("conv_dbg", 128)
            with ops.range_status(st):
                ops.resblock_chain(blocks, x)
            assert int(st.item()) == 2, "an abandoned chain sets bit 1 (and nothing else)"
            ops.resblock_chain(blocks, x)                 # no caller's word: the per-device fallback word takes the report
            with pytest.raises(RuntimeError, match="abandoned"):
                ops.check_chain_status()
        finally:
            ops.set_option("conv_dbg", 0)
        ops.check_chain_status()                          # read and cleared
        # (ii) the same launch through the C ABI with a workspace of TWO buffers: ids 2..4 of the table need three
        lib = _lib.load()
        n, c, h, w = x.shape
        d = blocks[0][0].desc(n, h, w, c)
        L = 2 * len(blocks)
        words = lib.motif_conv2d_chain_ws_words(ctypes.byref(d), L)
        assert words > 0
        tab = ops._chain_table(blocks, ops.ACT_RELU, ops.ACT_NONE, x.device)
        out = torch.full((n, c, h, w), 7.0, device=dev())
        work = torch.zeros(2 * n * c * h * w, device=dev())
        ws = torch.empty(words, dtype=torch.int32, device=dev())
        st.zero_()
        d.status = ctypes.c_void_p(st.data_ptr())
        rc = lib.motif_conv2d_chain_fwd(ctypes.byref(d), L, ctypes.c_void_p(tab.data_ptr()), ops._p(x), ops._p(out), ops._p(work), work.numel(), ops._p(ws), ops._stream())
        assert rc == 0
        assert int(st.item()) == 4 and bool((out == 7.0).all()) and bool((work == 0).all()), "a bad table must stop the launch before it writes"
        st.zero_()
        with ops.range_status(st):
            assert torch.equal(ops.resblock_chain(blocks, x), ref)
        assert int(st.item()) == 0
    finally:
        ops.CONV_CHAIN, ops.CONV_CHAIN_MIN_TILES = saved


@pytest.mark.parametrize("cin, cout, h, w, n", [(529, 2, 12, 20, 1), (661, 2, 24, 40, 2), (597, 2, 48, 80, 2), (565, 2, 96, 160, 1), (300, 1, 20, 24, 1), (277, 2, 7, 256, 1),
                                                (128, 2, 90, 160, 2), (96, 1, 130, 132, 1)])
def test_conv_direct_deep_form_with_whole_rows_per_workgroup_gives_the_bits_of_the_quad_form(cin, cout, h, w, n):
    """The 16-slice deep direct form on small maps (PWC-Net's flow heads), round 6: a workgroup owns whole rows and takes the pixels left and
    right of a quad from the neighbouring lanes (six channels' loads in flight instead of three) -- against the quad-indexed form with loaded
    edge pixels (option conv_direct_quads = 1): the same multiply-adds in the same order, so equal bits; rows of 5 .. 64 quads, a last
    workgroup with fewer rows, one and two couts, the 8-slice form too (RAFT's 128 -> 2 flow head); and against fp64."""
    from motif_amd import ops
    from motif_amd.models.modules.layers import Conv2d
    m = Conv2d(cin, cout, 3, 1, 1)
    with torch.no_grad():
        m.weight.copy_(rnd(*m.weight.shape, seed=1, scale=1.0 / math.sqrt(cin * 9)))
        m.bias.copy_(rnd(cout, seed=2, scale=0.1))
    x = rnd(n, cin, h, w, seed=3)
    y = F.conv2d(x.double(), m.weight.double(), m.bias.double(), 1, 1)
    m, xd = m.to(dev()), x.to(dev())
    rows = m(xd)
    ops.set_option("conv_direct_quads", 1)
    try:
        quads = m(xd)
    finally:
        ops.set_option("conv_direct_quads", 0)
    assert torch.equal(rows, quads)
    assert float((rows.double().cpu() - y).abs().max()) < 2e-6 * float(y.abs().max()) * max(1.0, math.sqrt(cin * 9 / 1800.0))


def test_conv_direct_deep_form_gives_an_image_the_same_bits_alone_and_in_a_batch():
    """the 16- / 8-slice choice of the deep direct form (the number of terms of its fixed-order partial-sum reduction) was made from
    N x workgroups-per-image, so a PWC-Net flow head gave a pair other bits in a batch of 9 than alone.  It is decided per image now."""
    from motif_amd import ops
    from motif_amd.models.modules.layers import Conv2d
    m = Conv2d(597, 2, 3, 1, 1)
    with torch.no_grad():
        m.weight.copy_(rnd(*m.weight.shape, seed=1, scale=1.0 / math.sqrt(597 * 9)))
        m.bias.copy_(rnd(2, seed=2, scale=0.1))
    m = m.to(dev())
    x = rnd(9, 597, 96, 160, seed=3).to(dev())          # 60 workgroups per image: 9 images crossed the old threshold of 512
    batch = m(x)
    for i in (0, 4, 8):
        assert torch.equal(m(x[i:i + 1])[0], batch[i])
    y = F.conv2d(x[:1].double().cpu(), m.weight.double().cpu(), m.bias.double().cpu(), 1, 1)
    assert float((batch[:1].double().cpu() - y).abs().max()) < 2e-6 * float(y.abs().max()) * math.sqrt(597 * 9 / 1800.0)


def test_conv_pw_guard_bands_channels_past_cin_and_couts_past_cout_touch_nothing(keep_mma):
    """conv_pw.hip relies on the buffer range check for channels past Cin (a ragged last 16-channel step), couts past
    Cout (a partial cout tile) and the masked lanes of a ragged pixel group.  The plane offsets are therefore part of the VECTOR
    offset (the scalar offset operand is outside the check).  Guard bands: NaN planes right behind every image's input and residual
    channels, canary planes right behind every image's output channels, Cin % 8 != 0, Cout % 32 != 0, N > 1, odd plane size."""
    from motif_amd import ops
    from motif_amd.models.modules.layers import Conv2d
    n, cin, cout, H, W, pad_planes = 2, 20, 40, 19, 33, 4
    m = Conv2d(cin, cout, 1, 1, 0)
    with torch.no_grad():
        m.weight.copy_(rnd(*m.weight.shape, seed=1, scale=0.2))
        m.bias.copy_(rnd(cout, seed=2, scale=0.1))
    x, res = rnd(n, cin, H, W, seed=3), rnd(n, cout, H, W, seed=4)
    ref = F.relu(F.conv2d(x.double(), m.weight.double(), m.bias.double()) + res.double())
    m = m.to(dev())
    xb = torch.full((n, cin + pad_planes, H, W), float("nan"), device=dev()); xb[:, :cin] = x.to(dev())
    rb = torch.full((n, cout + pad_planes, H, W), float("nan"), device=dev()); rb[:, :cout] = res.to(dev())
    ob = torch.full((n, cout + pad_planes, H, W), 123.0, device=dev())
    ops.set_conv_mma(ops.MMA_F16X2)
    m(xb[:, :cin], act=ops.ACT_RELU, res=rb[:, :cout], res_mode=1, out=ob[:, :cout])
    assert bool((ob[:, cout:] == 123.0).all()), "a cout past Cout was stored"
    assert bool(torch.isfinite(ob[:, :cout]).all()), "a channel past Cin / a residual plane past Cout was read"
    close(ob[:, :cout], ref.float(), 2e-5, 2e-5, "guard-banded 1x1 layer")


def test_conv_two_part_form_is_loud_outside_fp16_range_and_three_part_form_is_not(keep_mma):
    """mma = 7 (two fp16 parts) has fp16's range: a transformed activation beyond 65504 must give inf / NaN in the outputs it
    touches -- never a silently clamped finite value -- and leave every other output untouched; mma = 6 (three bf16 parts, fp32's
    exponent range) computes the same layer correctly.  include/motif_hip.h, MotifConvDesc.mma."""
    from motif_amd import ops
    from motif_amd.models.modules.layers import Conv2d
    m = Conv2d(64, 64, 3, 1, 1)
    with torch.no_grad():
        m.weight.copy_(rnd(*m.weight.shape, seed=1, scale=1.0 / 24))
        m.bias.copy_(rnd(64, seed=2, scale=0.1))
    x = rnd(1, 64, 32, 64, seed=3)
    x[0, 5, 10, 20] = 1.0e5
    ref = F.conv2d(x.double(), m.weight.double(), m.bias.double(), 1, 1)
    m = m.to(dev())
    try:
        ops.set_option("conv_engine", 5)
        ops.set_conv_mma(ops.MMA_BF16X3)
        o6 = m(x.to(dev())).cpu()
        ops.set_conv_mma(ops.MMA_F16X2)
        o7 = m(x.to(dev())).cpu()
    finally:
        ops.set_option("conv_engine", 0)
    close(o6, ref.float(), 2e-2, 1e-6, "three bf16 parts with a 1e5 activation")            # fp32 rounding of sums of magnitude 4e3
    bad = ~torch.isfinite(o7)
    assert bad.any(), "an out-of-range activation must not pass silently"
    ys, xs = bad.nonzero()[:, 2], bad.nonzero()[:, 3]
    assert int(ys.min()) >= 8 and int(ys.max()) <= 12 and int(xs.min()) >= 19 and int(xs.max()) <= 21, "only the outputs the value reaches (3 columns; the rows its Winograd row pairs cover)"
    near = torch.zeros_like(bad)
    near[:, :, 8:13, 19:22] = True
    close(o7[~near], ref.float()[~near], 2e-5, 2e-5, "outputs the value does not reach")


def _dcn_fp64(x, weight, bias, offset, mask, dg):
    """DCNv2 forward (3x3, pad 1) from its definition in fp64: tap k of output pixel p samples the input bilinearly at
    p - 1 + k + offset_k(p) (zero outside the image), times mask_k(p); then the dense contraction (dcn_v2_im2col_cuda.cu:125-194)."""
    x, weight, bias, offset, mask = (t.double() for t in (x, weight, bias, offset, mask))
    B, C, H, W = x.shape
    co = weight.shape[0]
    ys = torch.arange(H, dtype=torch.float64).view(1, H, 1)
    xs = torch.arange(W, dtype=torch.float64).view(1, 1, W)
    out = bias.view(1, co, 1, 1).expand(B, co, H, W).clone()
    cg = C // dg
    for g in range(dg):
        xg = x[:, g * cg:(g + 1) * cg]
        for k in range(9):
            py = ys - 1 + k // 3 + offset[:, g * 18 + 2 * k]
            px = xs - 1 + k % 3 + offset[:, g * 18 + 2 * k + 1]
            grid = torch.stack((2.0 * px / (W - 1) - 1.0, 2.0 * py / (H - 1) - 1.0), -1)
            smp = F.grid_sample(xg, grid, mode="bilinear", padding_mode="zeros", align_corners=True) * mask[:, g * 9 + k].unsqueeze(1)
            out += torch.einsum("oc,bchw->bohw", weight[:, g * cg:(g + 1) * cg, k // 3, k % 3], smp)
    return out


@pytest.mark.parametrize("ws", [1.0 / 24, 1e-2], ids=["w1/24", "w1e-2"])
@pytest.mark.parametrize("xs", [1.0, 1e-2, 1e-3, 1e-4], ids=["x1", "x1e-2", "x1e-3", "x1e-4"])
@pytest.mark.parametrize("layer", ["wino3x3", "pw1x1", "dcn", "ig16_s2", "ig16_7x7s2", "ig16_dil4", "ig16_narrow"])
def test_two_part_fp16_form_is_fp32_equivalent_at_every_activation_scale(layer, xs, ws, keep_mma):
    """the two-part fp16 form must not depend on the activations being of O(1).  The low activation part is stored
    times 2^11 (a normal fp16 number whenever the high part is one) and meets 2^-11 x the high weight part -- conv_wino.hip,
    conv_pw.hip, the DCN window kernel.  Against fp64, for activation scales 1 .. 1e-4 and weight scales 1/24, 1e-2, the error of
    mma = 7 stays within the bound the other split kernels are held to: 1.25 x the fp32-MFMA engine's (+ 1e-7 of the output scale)."""
    from motif_amd import ops
    from motif_amd.models.modules.layers import Conv2d
    modes = (ops.MMA_FP32, ops.MMA_BF16X3, ops.MMA_F16X2)
    err = {}
    if layer == "dcn":
        B, C, H, W, dg = 1, 64, 24, 40, 8
        x = rnd(B, C, H, W, seed=1, scale=xs)
        w = rnd(64, C, 3, 3, seed=2, scale=ws)
        bias = rnd(64, seed=3, scale=2.0 * xs * ws)
        off = rnd(B, 2 * dg * 9, H, W, seed=4, scale=3.0)
        mask = torch.sigmoid(rnd(B, dg * 9, H, W, seed=5, scale=2.0))
        ref = _dcn_fp64(x, w, bias, off, mask, dg)
        plan = ops.DcnPlan(w.to(dev()), bias.to(dev()))
        om = torch.cat([off, mask], 1).to(dev())
        for mode in modes:
            ops.set_conv_mma(mode)
            err[mode] = float((ops.dcn_v2_multi([plan], [x.to(dev())], [om], dg)[0].double().cpu() - ref).abs().max())
    else:
        # (cin, cout, k, stride, pad, dil); the ig16_* layers are conv_ig16.hip's (round 6): stride 2, a 7x7 stem, dilation, 24 -> 24 narrow
        cin, cout, k, st, pd, dl = {"wino3x3": (64, 64, 3, 1, 1, 1), "pw1x1": (128, 64, 1, 1, 0, 1), "ig16_s2": (64, 64, 3, 2, 1, 1), "ig16_7x7s2": (3, 32, 7, 2, 3, 1),
                                    "ig16_dil4": (96, 128, 3, 1, 4, 4), "ig16_narrow": (24, 24, 3, 1, 1, 1)}[layer]
        m = Conv2d(cin, cout, k, st, pd, dl)
        with torch.no_grad():
            m.weight.copy_(rnd(*m.weight.shape, seed=1, scale=ws))
            m.bias.copy_(rnd(cout, seed=2, scale=2.0 * xs * ws))
        x = rnd(2, cin, 45, 80, seed=3, scale=xs)
        ref = F.conv2d(x.double(), m.weight.double(), m.bias.double(), st, pd, dl)
        m = m.to(dev())
        outs = {}
        try:
            if layer.startswith("ig16"):
                ops.set_option("conv_engine", 7)             # conv_ig16.hip wherever it fits (the dispatch gives it only the shapes it wins on)
            for mode in modes:
                ops.set_conv_mma(mode)
                outs[mode] = m(x.to(dev())).double().cpu()
                err[mode] = float((outs[mode] - ref).abs().max())
        finally:
            ops.set_option("conv_engine", 0)
        assert not torch.equal(outs[ops.MMA_F16X2], outs[ops.MMA_FP32]), "mma = 7 ran its own kernel"
        if not layer.startswith("ig16"):                 # (mma = 6 runs these shapes on the fp32 engine: conv_ig16.hip has the two-part form only)
            assert not torch.equal(outs[ops.MMA_F16X2], outs[ops.MMA_BF16X3])
    scale = float(ref.abs().max())
    assert err[ops.MMA_FP32] < 3e-6 * scale, (err, scale)
    assert err[ops.MMA_BF16X3] <= 1.25 * err[ops.MMA_FP32] + 1e-7 * scale, (err, scale)
    assert err[ops.MMA_F16X2] <= 1.25 * err[ops.MMA_FP32] + 1e-7 * scale, (err, scale)


IG16_CASES = [
    # cin, cout, k, stride, pad, dil, groups, pad_mode, H, W, N, c0 (two-source split, 0 = one source), act, res_mode, act_split
    (64, 64, 3, 2, 1, 1, 1, "zeros", 90, 160, 2, 0, "lrelu", 0, 0),          # the PCD / ZSM pyramid layers (Ours.py:141-144): VEC staging, one octet per chunk
    (64, 64, 3, 2, 1, 1, 1, "zeros", 45, 78, 1, 0, "none", 0, 0),            # W % 4 != 0: scalar staging plan, ragged tiles
    (3, 32, 7, 2, 3, 1, 1, "zeros", 128, 192, 2, 0, "relu", 0, 0),           # RAFT's stem: 3 real channels of the octet, 25 k-steps
    (2, 64, 7, 1, 3, 1, 1, "zeros", 24, 40, 2, 0, "relu", 0, 0),             # motion encoder convf1: one cout tile per block (the 7x7 fragments of two do not fit)
    (128, 128, 3, 1, 2, 2, 1, "zeros", 48, 64, 1, 0, "lrelu", 0, 0),         # PWC-Net's refiner, dilation 2 .. 8
    (128, 96, 3, 1, 8, 8, 1, "zeros", 48, 64, 1, 0, "lrelu", 0, 0),
    (96, 64, 3, 1, 16, 16, 1, "zeros", 48, 64, 1, 0, "lrelu", 0, 0),         # dilation 16: the patch of an octet does not fit -> stays on the fp32 engine (same answer)
    (64, 32, 3, 1, 1, 1, 1, "zeros", 23, 40, 2, 0, "relu", 0, 0),            # 17 .. 32 couts with a short reduction (convf2)
    (24, 24, 3, 2, 1, 1, 1, "zeros", 90, 160, 2, 0, "none", 2, 0),           # bottleneck conv2 with stride 2, residual after the activation
    (32, 64, 1, 2, 0, 1, 1, "zeros", 40, 64, 2, 0, "none", 0, 0),            # down-sampling 1x1 stride 2: one tap per octet (half of every k-step empty)
    (160, 96, 3, 2, 1, 1, 1, "reflect", 20, 28, 1, 64, "tanh", 4, 0),        # two sources (C0 = 64), reflect padding, stride 2, multiplicative residual
    (14, 64, 3, 1, 1, 1, 2, "zeros", 32, 48, 2, 0, "lrelu", 1, 0),           # groups = 2 with 7 channels each (flow_process[0]), residual before the activation
    (40, 48, 5, 1, 2, 1, 1, "zeros", 21, 36, 1, 0, "sigmoid", 0, 24),        # 5x5, activation split on a cout boundary
    (200, 160, 1, 1, 0, 1, 1, "zeros", 19, 33, 1, 0, "relu", 0, 0),          # a 1x1 wider than conv_pw.hip takes
]


@pytest.mark.parametrize("case", IG16_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv_ig16_generic_layers_on_the_fp16_cores_match_fp64_and_the_fp32_engine(case, keep_mma):
    """Round 6: conv_ig16.hip -- stride 2, 7x7, dilated, narrow / wide, grouped, two-source, reflect-padded layers as an
    implicit GEMM on the fp16 matrix cores with the two-part arithmetic.  Against fp64 with the bound every split kernel is held to
    (error <= 1.25 x the fp32-MFMA engine's + 1e-7 of the output scale), and the launch must really have taken the new kernel (option
    conv_engine = 6 = "fp32 engine for these shapes" gives other bits) except where the shape does not fit it."""
    from motif_amd import ops
    from motif_amd.models.modules.layers import Conv2d
    cin, cout, k, st, pd, dl, groups, pm, H, W, N, c0, actn, rm, asplit = case
    acts = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU, "tanh": ops.ACT_TANH, "sigmoid": ops.ACT_SIGMOID}
    fact = {"none": lambda v: v, "relu": F.relu, "lrelu": lambda v: F.leaky_relu(v, 0.1), "tanh": torch.tanh, "sigmoid": torch.sigmoid}[actn]
    m = Conv2d(cin, cout, k, st, pd, dl, groups, True, pm)
    with torch.no_grad():
        m.weight.copy_(rnd(*m.weight.shape, seed=1, scale=1.0 / math.sqrt(cin * k * k / groups)))
        m.bias.copy_(rnd(cout, seed=2, scale=0.1))
    x = rnd(N, cin, H, W, seed=3)
    xp = F.pad(x.double(), (pd,) * 4, mode="reflect") if pm == "reflect" else x.double()
    y = F.conv2d(xp, m.weight.double(), m.bias.double(), st, 0 if pm == "reflect" else pd, dl, groups)
    res = rnd(*y.shape, seed=4)
    r = res.double()
    if rm == 1: y = y + r
    y = torch.cat([fact(y[:, :asplit]), F.relu(y[:, asplit:])], 1) if asplit else fact(y)
    if rm == 2: y = y + r
    elif rm == 4: y = y * r
    m = m.to(dev())
    xd, rd = x.to(dev()), res.to(dev())
    args = (xd[:, :c0].contiguous(), xd[:, c0:].contiguous()) if c0 else (xd, None)
    kw = dict(act=acts[actn], res=rd if rm else None, res_mode=rm)
    if asplit: kw.update(act2=ops.ACT_RELU, act_split=asplit)
    outs = {}
    try:
        ops.set_conv_mma(ops.MMA_F16X2)
        ops.set_option("conv_engine", 7)                 # the new kernel wherever the shape FITS (the default dispatch: only where it also pays)
        outs["f16x2"] = m(*args, **kw).double().cpu()
        ops.set_option("conv_engine", 6)
        outs["forced_fp32_engine"] = m(*args, **kw).double().cpu()
        ops.set_option("conv_engine", 0)
        ops.set_conv_mma(ops.MMA_FP32)
        outs["fp32"] = m(*args, **kw).double().cpu()
    finally:
        ops.set_option("conv_engine", 0)
    scale = float(y.abs().max())
    err = {kk: float((v - y).abs().max()) for kk, v in outs.items()}
    assert err["fp32"] < 3e-6 * scale * max(1.0, math.sqrt(cin * k * k / groups / 600.0)), (err, scale)
    assert err["f16x2"] <= 1.25 * err["fp32"] + 1e-7 * scale, (err, scale)
    fits = not (dl == 16)                                # an octet's patch at dilation 16 is 20 480 floats: beyond one LDS buffer
    assert torch.equal(outs["f16x2"], outs["forced_fp32_engine"]) != fits, "the launch did not take the kernel it should: %s" % (err,)
    # the default dispatch: the layers named in conv_ig16.hip's rule (>= 24 channels per group and stride / dilation / width) take it, the others do not
    pays = cin // groups >= 24 and (cin // groups * k * k >= 200 or cin // groups >= 64) and (st > 1 or dl > 1 or cin // groups >= 64 or cout // groups >= 48)
    ops.set_conv_mma(ops.MMA_F16X2)
    default = m(*args, **kw).double().cpu()
    assert torch.equal(default, outs["f16x2"] if (pays and fits) else outs["forced_fp32_engine"])


def test_range_status_word_is_set_by_the_kernel_that_meets_an_out_of_range_operand(keep_mma):
    """include/motif_hip.h "Range status word": every kernel of the two-part fp16 form ORs bit 0 into the caller's word when an
    operand of its own launch left fp16's range -- at the source, whatever later stages do with the value -- and leaves the word
    alone on in-range data, with no word given, and under mma = 6."""
    from motif_amd import ops
    from motif_amd.models.modules.layers import Conv2d
    word = torch.zeros(1, dtype=torch.int32, device=dev())

    def fired(fn):
        word.zero_()
        with ops.range_status(word):
            out = fn()
        return int(word.item()), out

    # 3x3 (conv_wino.hip): one activation of 1e5 in a corner tile of a ragged map; 4e4 overflows through the row transform (d1 + d2)
    m3 = Conv2d(64, 64, 3, 1, 1)
    m1 = Conv2d(128, 40, 1, 1, 0)                         # conv_pw.hip, partial second cout tile
    with torch.no_grad():
        m3.weight.copy_(rnd(*m3.weight.shape, seed=1, scale=1.0 / 24)); m3.bias.zero_()
        m1.weight.copy_(rnd(*m1.weight.shape, seed=2, scale=0.05)); m1.bias.zero_()
    m3, m1 = m3.to(dev()), m1.to(dev())
    x3 = rnd(2, 64, 37, 52, seed=3).to(dev())
    x1 = rnd(2, 128, 19, 33, seed=4).to(dev())
    ops.set_conv_mma(ops.MMA_F16X2)
    assert fired(lambda: m3(x3))[0] == 0 and fired(lambda: m1(x1))[0] == 0
    for val, where in ((1.0e5, (1, 63, 36, 51)), (-7.0e4, (0, 0, 0, 0)), (float("inf"), (1, 17, 20, 31))):
        xb = x3.clone(); xb[where] = val
        flag, out = fired(lambda: m3(xb))
        assert flag == 1 and not bool(torch.isfinite(out).all()), (val, where)
    xb = x3.clone(); xb[0, 5, 10, 20] = 4.0e4; xb[0, 5, 11, 20] = 4.0e4          # each fits fp16, their Winograd sum d1 + d2 does not
    assert fired(lambda: m3(xb))[0] == 1
    xb = x1.clone(); xb[1, 127, 18, 32] = 7.0e4
    assert fired(lambda: m1(xb))[0] == 1
    # conv_ig16.hip (round 6): a stride-2 3x3 layer; the out-of-range value sits where only ONE output pixel's window covers it
    ms = Conv2d(64, 64, 3, 2, 1)
    with torch.no_grad():
        ms.weight.copy_(rnd(*ms.weight.shape, seed=9, scale=1.0 / 24)); ms.bias.zero_()
    ms = ms.to(dev())
    assert fired(lambda: ms(x3))[0] == 0
    for val, where in ((1.0e5, (1, 63, 36, 51)), (-7.0e4, (0, 0, 0, 0)), (float("inf"), (1, 17, 21, 31))):
        xb = x3.clone(); xb[where] = val
        flag, out = fired(lambda: ms(xb))
        assert flag == 1 and not bool(torch.isfinite(out).all()), (val, where)
    # no word: nothing to write, nothing breaks; mma = 6 computes the same layers in range
    xb = x3.clone(); xb[0, 3, 3, 3] = 1.0e5
    m3(xb)
    ops.set_conv_mma(ops.MMA_BF16X3)
    flag, out = fired(lambda: m3(xb))
    assert flag == 0 and bool(torch.isfinite(out).all())
    # DCN window kernel
    ops.set_conv_mma(ops.MMA_F16X2)
    B, C, H, W, dg = 1, 64, 24, 40, 8
    xd = rnd(B, C, H, W, seed=5).to(dev())
    plan = ops.DcnPlan(rnd(64, C, 3, 3, seed=6, scale=0.05).to(dev()), torch.zeros(64, device=dev()))
    om = torch.cat([rnd(B, 2 * dg * 9, H, W, seed=7, scale=2.0), torch.sigmoid(rnd(B, dg * 9, H, W, seed=8, scale=2.0)) * 0.5 + 0.5], 1).to(dev())
    assert fired(lambda: ops.dcn_v2_multi([plan], [xd], [om], dg))[0] == 0
    xb = xd.clone(); xb[0, 9, 12, 17] = 1.0e6           # blended and masked it still exceeds fp16's range somewhere
    assert fired(lambda: ops.dcn_v2_multi([plan], [xb], [om], dg))[0] == 1


# ------------------------------------------------------------------------------------------- DCNv2
def test_dcn_matches_kernel_text_restatement():
    from oracle import native
    from motif_amd import ops
    B, C, H, W, dg = 2, 64, 23, 37, 8
    x = rnd(B, C, H, W, seed=1)
    w = rnd(64, C, 3, 3, seed=2, scale=0.05)
    bias = rnd(64, seed=3, scale=0.1)
    off = rnd(B, 2 * dg * 9, H, W, seed=4, scale=3.0)
    mask = torch.sigmoid(rnd(B, dg * 9, H, W, seed=5, scale=2.0))
    ref = native.dcn_v2_forward(x, w, bias, off, mask, 3, 3, 1, 1, 1, 1, 1, 1, dg)
    out = ops.dcn_v2_raw(x.to(dev()), off.to(dev()), mask.to(dev()), w.to(dev()), bias.to(dev()), 3, 3, 1, 1, 1, dg)
    close(out, ref, 3e-5, 3e-5, "dcn")


def test_dcn_zero_offset_identity_known_answer():
    """The reference's own check (models/modules/DCNv2/test.py:32-67): zero offsets, mask 0.5, identity
    centre-tap weights => 2*output == input."""
    from motif_amd import ops
    N, C, H, W, dg = 2, 4, 8, 8, 2
    w = torch.zeros(C, C, 3, 3)
    for p in range(C):
        w[p, p, 1, 1] = 1.0
    x = torch.randn(N, C, H, W, generator=torch.Generator().manual_seed(0))
    off = torch.zeros(N, dg * 18, H, W)
    mask = torch.full((N, dg * 9, H, W), 0.5)
    out = ops.dcn_v2_raw(x.to(dev()), off.to(dev()), mask.to(dev()), w.to(dev()), torch.zeros(C, device=dev()), 3, 3, 1, 1, 1, dg)
    assert float((x - 2 * out.cpu()).abs().max()) < 1e-10


def test_dcn_sep_module_fused_offset_mask(engine):
    from oracle.motif_ref import DcnSep
    from motif_amd.models.modules.DCNv2.dcn_v2 import DCN_sep
    ref = DcnSep(64, 8)
    with torch.no_grad():
        for i, p in enumerate(ref.parameters()):
            p.copy_(rnd(*p.shape, seed=10 + i, scale=0.05))
    mine = DCN_sep(64, 64, 3, 1, 1, 1, 8)
    mine.load_state_dict(ref.state_dict())
    x, fea = rnd(1, 64, 20, 33, seed=1), rnd(1, 64, 20, 33, seed=2, scale=2.0)
    with torch.no_grad():
        r = ref(x, fea)
    out = mine.to(dev())(x.to(dev()), fea.to(dev()))
    close(out, r, 5e-5, 5e-5, "DCN_sep")


# ------------------------------------------------------------------------------------------- splat
def _flow(n, h, w, seed, mag=4.0):
    f = rnd(n, 2, h, w, seed=seed, scale=mag)
    f[:, :, :2, :] *= 10.0  # throw some sources far outside the frame
    return f


def test_splat_operator_form_vs_kernel_text():
    from oracle import native
    from motif_amd import ops
    n, c, h, w = 3, 5, 37, 53
    src, flow, z = rnd(n, c, h, w, seed=1), _flow(n, h, w, 2), rnd(n, 1, h, w, seed=3)
    ez = z.exp()
    ref = native.splat(torch.cat([src * ez, ez], 1), flow, "sum")
    o = ops.splat(src.to(dev()), flow.to(dev()), z.to(dev()), want=("sum", "norm", "max", "cnt"))
    close(o["sum"], ref[:, :-1], 2e-5, 1e-5, "splat sum")
    close(o["norm"], ref[:, -1:], 2e-5, 1e-5, "splat norm")
    cnt = native.splat(torch.ones(n, 1, h, w), flow, "count")
    assert torch.equal(o["cnt"].cpu(), cnt), "hit count must be bit-exact (integer valued)"
    # max: exact given the same e^z -> feed e^z as the plain input (z=None path)
    mx = ops.splat(ez.to(dev()), flow.to(dev()), None, want=("max",))["max"]
    assert torch.equal(mx.cpu(), native.splat(ez, flow, "max"))


def test_splat_max_with_values_above_one_and_modules():
    from oracle import native
    from motif_amd.models.softsplat_cp import Softsplat
    from motif_amd.models.softsplat_count_cp import Softsplat_Count
    from motif_amd.models.softsplat_max_cp import Softsplat_Max
    n, h, w = 2, 24, 40
    img, flow = rnd(n, 1, h, w, seed=1).abs() * 5.0, _flow(n, h, w, 2, 1.5)
    assert torch.equal(Softsplat_Max()(img.to(dev()), flow.to(dev())).cpu(), native.splat(img, flow, "max"))
    assert torch.equal(Softsplat_Count()(img.to(dev()), flow.to(dev())).cpu(), native.splat(torch.ones_like(img), flow, "count"))
    feat, z = rnd(n, 7, h, w, seed=4), rnd(n, 1, h, w, seed=5)
    out, norm = Softsplat()(feat.to(dev()), flow.to(dev()), z.to(dev()))
    ref = native.splat(torch.cat([feat * z.exp(), z.exp()], 1), flow, "sum")
    close(out, ref[:, :-1], 2e-5, 1e-5)
    close(norm, ref[:, -1:], 2e-5, 1e-5)


def test_splat_empty_flow_is_identity_count_four_corners():
    from motif_amd import ops
    flow = torch.zeros(1, 2, 16, 64, device=dev())
    o = ops.splat(None, flow, None, want=("cnt",))
    # zero flow: weights (1,0,0,0) but all four corners are "touched" where in bounds
    cnt = o["cnt"].cpu()[0, 0]
    assert cnt[5, 5] == 4 and cnt[0, 0] == 1 and cnt[0, 5] == 2


# ------------------------------------------------------------------------------------------- resampling
def test_resize_with_raft_normalisation_and_into_a_slice_is_bit_exact():
    """motif_resize_bilinear post = 1: the HR frames come out as RAFT's normalised input, 2 * ((v * 255) / 255) - 1 with the roundings of
    the reference's four element-wise operations (Ours.py:544, raft.py:90-91); `out=`: the result lands in a slice of a wider tensor."""
    from motif_amd import ops
    x = torch.rand(4, 3, 36, 64, generator=torch.Generator().manual_seed(3))
    plain = ops.resize_bilinear(x.to(dev()), (144, 256), False)
    # the four operations on the HOST: torch on a GPU divides by a scalar as a multiplication by its rounded reciprocal, the kernel
    # (like the CPU reference the goldens come from) divides
    want = 2 * ((plain.cpu() * 255.0) / 255.0) - 1.0
    got = ops.resize_bilinear(x.to(dev()), (144, 256), False, raft_norm=True)
    assert torch.equal(got.cpu(), want)
    buf = torch.full((6, 3, 18, 32), 7.0, device=dev())
    ops.resize_bilinear(x.to(dev()), (18, 32), False, 0.5, out=buf[1:5])
    assert torch.equal(buf[1:5], ops.resize_bilinear(x.to(dev()), (18, 32), False, 0.5)) and bool((buf[0] == 7.0).all()) and bool((buf[5] == 7.0).all())


@pytest.mark.parametrize("align", [False, True])
@pytest.mark.parametrize("shape", [((18, 32), (72, 128)), ((72, 128), (18, 32)), ((9, 16), (18, 32)), ((45, 80), (90, 160)), ((7, 10), (14, 20)), ((16, 24), (128, 192)), ((23, 31), (47, 50))])
def test_resize_bilinear(shape, align):
    from motif_amd import ops
    (h, w), (ho, wo) = shape
    x = rnd(3, 2, h, w, seed=1)
    ref = F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=align) * 0.25
    close(ops.resize_bilinear(x.to(dev()), (ho, wo), align, 0.25), ref, 2e-6, 0, "resize")


@pytest.mark.parametrize("mode", ["f16x2", "fp32"])
@pytest.mark.parametrize("h, w", [(45, 80), (90, 160)])
def test_an_image_has_the_same_bits_alone_and_in_a_batch_for_every_operator_of_the_alignment(mode, h, w, keep_mma):
    """Batch invariance in the default and the fp32 arithmetic: the two-source 3x3 layer, the plain 3x3 layer, the offset layer
    + deformable convolution and the x2 resize of the PCD alignment (Ours.py:107-172) on a batch of three images against each image alone --
    what the bit-identity of row-tiled and per-timestamp multi-GPU rendering rests on.  (`bf16x3` picks between two 3x3 kernels by the
    launch's tile count and is NOT batch-invariant on these map sizes: documented, not tested here.)"""
    from motif_amd import ops
    from motif_amd.models.modules.Ours import PCD_Align, convm, dcnm, up2m, LRELU
    from motif_amd.utils.synth_weights import fill_state_dict
    ops.set_mma(mode)
    m = fill_state_dict(PCD_Align(64, 8, use_time=False)).to(dev()).eval()
    a, b = rnd(3, 64, h, w, seed=1).to(dev()), rnd(3, 64, h, w, seed=2).to(dev())
    layers = {
        "3x3 128->64, two sources": lambda x, y: convm([m.L1_offset_conv1_1, m.L1_offset_conv1_2], [x, y], [y, x], act=LRELU),
        "3x3 64->64": lambda x, y: convm([m.L1_offset_conv3_1, m.L1_offset_conv3_2], [x, y], act=LRELU),
        "offset layer 64->216 + deformable convolution": lambda x, y: dcnm([m.L1_dcnpack_1, m.L1_dcnpack_2], [x, y], [y, x], LRELU),
        "x2 resize": lambda x, y: up2m(torch.stack([x, y]), 2.0),
    }
    with torch.no_grad():
        for name, fn in layers.items():
            full = fn(a, b)
            for i in range(3):
                assert torch.equal(full[:, i:i + 1], fn(a[i:i + 1], b[i:i + 1])), "%s: image %d differs from itself alone (%s)" % (name, i, mode)


@pytest.mark.parametrize("n, c, h, w", [(8, 64, 45, 80), (2, 3, 9, 16), (1, 1, 2, 4), (1, 2, 7, 12), (3, 5, 90, 160)])
def test_resize_x2_wide_form_gives_the_bits_of_the_narrow_form_and_matches_torch(n, c, h, w):
    """x2 upsampling with 8 columns x 2 rows per thread (round 6: 9 loads per 16 outputs instead of 32) against the 4-column form it replaces
    (option resize_narrow = 1): the same expression of the same operands per output, so the bits must be equal -- first / last rows and
    columns (clamped windows), the two-row minimum, the scale factor and RAFT's normalisation included; and against torch."""
    from motif_amd import ops
    x = rnd(n, c, h, w, seed=11, scale=3.0)
    xd = x.to(dev())
    wide = ops.resize_bilinear(xd, (2 * h, 2 * w), False, 0.25)
    wide_norm = ops.resize_bilinear(xd, (2 * h, 2 * w), False, raft_norm=True)
    ops.set_option("resize_narrow", 1)
    try:
        narrow = ops.resize_bilinear(xd, (2 * h, 2 * w), False, 0.25)
        narrow_norm = ops.resize_bilinear(xd, (2 * h, 2 * w), False, raft_norm=True)
    finally:
        ops.set_option("resize_narrow", 0)
    assert torch.equal(wide, narrow) and torch.equal(wide_norm, narrow_norm)
    close(wide, F.interpolate(x, size=(2 * h, 2 * w), mode="bilinear", align_corners=False) * 0.25, 2e-6, 0, "resize x2")


@pytest.mark.parametrize("shape", [(2, 32, 360, 640), (2, 96, 90, 160), (1, 5, 37, 53)])
def test_instance_norm_all_modes_large_planes(shape):
    """The split-plane instance norm (moments in fp64 over slices, 16-byte accesses where the plane allows) against torch, with
    the fused relu / residual epilogues; the odd plane takes the 4-byte path."""
    from motif_amd import ops
    n, c, h, w = shape
    x, res = rnd(n, c, h, w, seed=5, scale=2.0) + 0.3, rnd(n, c, h, w, seed=6)
    ref = F.instance_norm(x, eps=1e-5)
    xd, rd = x.to(dev()), res.to(dev())
    close(ops.instance_norm(xd, 0), ref, 2e-5, 1e-5, "mode 0")
    close(ops.instance_norm(xd, 1), F.relu(ref), 2e-5, 1e-5, "mode 1")
    close(ops.instance_norm(xd, 2, res=rd), F.relu(F.relu(ref) + res), 2e-5, 1e-5, "mode 2")


def test_instance_norm_with_statistics_over_row_bands():
    """motif_instance_norm_moments / _apply (statistics of a row-tiled clip all-reduced over ranks): moments over two disjoint row
    ranges, summed, then applied to the whole plane == F.instance_norm of the plane, for every epilogue mode; a halo row outside the
    owned range must not enter the statistics."""
    from motif_amd import ops
    x = rnd(2, 5, 40, 24, seed=1, scale=2.0) + 0.7
    res = rnd(2, 5, 40, 24, seed=2)
    ref = F.instance_norm(x, eps=1e-5)
    xd, rd = x.to(dev()), res.to(dev())
    parts = []

    def collect(t):
        parts.append(t.clone())
        return t
    ops.instance_norm_synced(xd, (0, 16), collect, 0)
    ops.instance_norm_synced(xd, (16, 40), collect, 0)
    total = parts[0] + parts[1]
    assert float(total[0, 2]) == 40 * 24

    def use_total(t):
        t.copy_(total)
        return t
    close(ops.instance_norm_synced(xd, (0, 16), use_total, 0), ref, 2e-5, 1e-5, "mode 0")
    close(ops.instance_norm_synced(xd, (16, 40), use_total, 1), F.relu(ref), 2e-5, 1e-5, "mode 1")
    close(ops.instance_norm_synced(xd, (3, 9), use_total, 2, res=rd), F.relu(F.relu(ref) + res), 2e-5, 1e-5, "mode 2")
    close(ops.instance_norm(xd, 0), ref, 2e-5, 1e-5, "single-process kernel")
    sub = F.instance_norm(x[:, :, 8:24], eps=1e-5)                     # statistics of rows 8..23 only, applied to those rows
    got = ops.instance_norm_synced(xd, (8, 24), lambda t: t, 0)[:, :, 8:24]
    close(got, sub, 2e-5, 1e-5, "own rows only")


def test_backwarp_and_reliability_maps():
    from oracle.motif_ref import MotifRef, back_warp
    from motif_amd import ops
    B, H, W = 2, 24, 40
    img, flow = rnd(4 * B, 3, H, W, seed=1), rnd(4 * B, 2, H, W, seed=2, scale=3.0)
    close(ops.backwarp(img.to(dev()), flow.to(dev())), back_warp(img, flow), 3e-6, 0, "backwarp")
    # reliability maps against the restatement's motion_and_reliability internals
    fr = torch.rand(B, 4, 3, H, W, generator=torch.Generator().manual_seed(3))
    fl = flow.clone()
    fl[:B] = 0
    fl[3 * B:] = 0
    g = torch.tensor([[1 / 16, 1 / 8, 1 / 16], [1 / 8, 1 / 4, 1 / 8], [1 / 16, 1 / 8, 1 / 16]]).reshape(1, 1, 1, 3, 3)
    fr0, fr1 = fr[:, 1], fr[:, 2]
    warped = back_warp(torch.cat([fr0, fr1, fr0, fr1], 0), fl)
    psi_photo = (torch.cat([fr0, fr0, fr1, fr1], 0) - warped).abs().mean(1)
    f4 = fl.reshape(4, B, 2, H, W)
    warped = back_warp(-torch.cat([f4[0], f4[2], f4[1], f4[3]], 0), fl)
    psi_flow = (fl - warped).abs().mean(1)
    sq, mean = torch.split(F.conv3d(F.pad(torch.cat([fl ** 2, fl], 1), (1, 1, 1, 1), mode="reflect").unsqueeze(1), g).squeeze(1), 2, dim=1)
    psi_var = (sq - mean ** 2).clip(1e-9, None).sqrt().mean(1)
    psies = torch.stack([psi_photo, psi_flow / 10.0, psi_var], 1)
    dur = torch.tensor([[0, 0], [0, 8], [8, 0], [8, 8]], dtype=torch.float32).unsqueeze(1)
    ff = torch.cat(((fl / 20.0).reshape(2, 2, B, -1, H, W).permute(0, 2, 1, 3, 4, 5).reshape(2 * B, 2, -1, H, W),
                    psies.reshape(2, 2, B, -1, H, W).permute(0, 2, 1, 3, 4, 5).reshape(2 * B, 2, -1, H, W),
                    dur.reshape(2, 4, 1, 1).unsqueeze(1).repeat(1, B, 1, H, W).reshape(2 * B, 2, 2, H, W) / 8.0), dim=2).reshape(2 * B, -1, H, W)
    frd = fr.to(dev())
    p, f = ops.reliability(frd[:, 1], frd[:, 2], fl.to(dev()), g.to(dev()), B, H, W)
    close(p, psies, 5e-6, 1e-5, "psies")
    close(f, ff, 5e-6, 1e-5, "flow_feat")


@pytest.mark.parametrize("n, c, h, w", [(2, 5, 24, 40), (1, 128, 24, 40), (2, 33, 12, 20), (1, 13, 7, 70)],
                         ids=["c5", "c128_level6", "ragged_channel_group", "ragged_row"])
def test_pwc_backward_warp(n, c, h, w):
    """PWCNet.py:146-177 against the oracle; the kernel spreads a pixel's channels over workgroups of 8 channels x 64 pixels (round 6), so a
    channel count that is not a multiple of 8 and a row that is not a multiple of 64 are cases of their own."""
    from oracle.pwc_ref import backward_warp
    from motif_amd import ops
    img, flow = rnd(n, c, h, w, seed=1), rnd(n, 2, h, w, seed=2, scale=6.0)
    close(ops.pwc_backward_warp(img.to(dev()), flow.to(dev())), backward_warp(img, flow), 3e-6, 0, "pwc warp")


# ------------------------------------------------------------------------------------------- norms / gates
def test_instance_norm_modes():
    from motif_amd import ops
    x, res = rnd(2, 8, 45, 80, seed=1) * 3 + 0.5, rnd(2, 8, 45, 80, seed=2)
    n = F.instance_norm(x)
    xd, rd = x.to(dev()), res.to(dev())
    close(ops.instance_norm(xd, 0), n, 3e-6, 1e-5)
    close(ops.instance_norm(xd, 1), F.relu(n), 3e-6, 1e-5)
    close(ops.instance_norm(xd, 2, res=rd), F.relu(res + F.relu(n)), 3e-6, 1e-5)


def test_pool_transpose_gates_axpby():
    from motif_amd import ops
    x = rnd(2, 6, 16, 24, seed=1)
    close(ops.avg_pool2(x.to(dev())), F.avg_pool2d(x, 2, stride=2), 1e-7)
    close(ops.nchw_to_nhwc(x.to(dev())), x.permute(0, 2, 3, 1), 0)
    x2 = rnd(1, 130, 7, 9, seed=2)
    close(ops.nchw_to_nhwc(x2.to(dev())), x2.permute(0, 2, 3, 1), 0)
    z, q, h = torch.sigmoid(rnd(2, 6, 16, 24, seed=2)), rnd(2, 6, 16, 24, seed=3), rnd(2, 6, 16, 24, seed=4)
    close(ops.gru_update(z.to(dev()), q.to(dev()), h.to(dev())), (1 - z) * h + z * q, 1e-6)
    cc, c = rnd(2, 16, 8, 12, seed=5, scale=3), rnd(2, 4, 8, 12, seed=6)
    i, f, o, g = torch.split(cc, 4, 1)
    cn = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    hn = torch.sigmoid(o) * torch.tanh(cn)
    h2, c2 = ops.lstm_gates(cc.to(dev()), c.to(dev()))
    close(h2, hn, 2e-6)
    close(c2, cn, 2e-6)
    close(ops.axpby(x.to(dev()), (x * 2).to(dev()), 1.0, -1.0), -x, 0)
    # ragged sizes: the 16-byte paths hand the last n % 4 values (or an odd plane) to the one-by-one code
    z, q, h = torch.sigmoid(rnd(1, 3, 7, 9, seed=12)), rnd(1, 3, 7, 9, seed=13), rnd(1, 3, 7, 9, seed=14)
    close(ops.gru_update(z.to(dev()), q.to(dev()), h.to(dev())), (1 - z) * h + z * q, 1e-6)
    xr = rnd(1, 3, 7, 9, seed=15)
    close(ops.axpby(xr.to(dev()), (xr * 2).to(dev()), 1.0, -1.0), -xr, 0)
    close(ops.axpby(xr.to(dev()), None, 0.5, 0.0), xr * 0.5, 0)
    # batch-strided destination (RAFT: flow = coords1 - coords0 written into two channels of the GRU input buffer)
    a, b = rnd(3, 2, 9, 13, seed=18), rnd(3, 2, 9, 13, seed=19)
    buf = torch.full((3, 7, 9, 13), 5.0, device=dev())
    got = ops.axpby_into(a.to(dev()), b.to(dev()), 1.0, -1.0, buf[:, 4:6])
    assert got.data_ptr() == buf[:, 4:6].data_ptr() and torch.equal(buf[:, 4:6].cpu(), a - b)
    assert bool((buf[:, :4] == 5.0).all()) and bool((buf[:, 6:] == 5.0).all())
    cc, c = rnd(2, 8, 5, 7, seed=16, scale=3), rnd(2, 2, 5, 7, seed=17)
    i, f, o, g = torch.split(cc, 2, 1)
    cn = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h2, c2 = ops.lstm_gates(cc.to(dev()), c.to(dev()))
    close(h2, torch.sigmoid(o) * torch.tanh(cn), 2e-6)
    close(c2, cn, 2e-6)
    pred = rnd(3, 3, 24, 40, seed=9, scale=0.3)
    for ratio in (4.0, 3.0, 2.5):
        want = pred[:, :2] * 20.0 * ratio / 20.0 / ratio
        assert torch.equal(ops.flow_roundtrip(pred.to(dev()), 20.0, ratio).cpu(), want), "flow round trip must keep the four roundings"


@pytest.mark.parametrize("n, cin, h, w", [(2, 37, 6, 10), (1, 597, 24, 40), (2, 70, 9, 130), (1, 64, 5, 7)], ids=["plain", "sliced_597", "sliced_ragged_70", "sliced_64"])
def test_deconv4x4s2(n, cin, h, w):
    """ConvTranspose2d(k 4, s 2, p 1) of PWC-Net's moduleUpflow / moduleUpfeat (PWCNet.py:117-125); >= 64 input channels take the sliced form
    (8 channel slices per workgroup, eight channels' loads in flight per step and a remainder loop: 597 = 8 x 75 - 3, 70 = 8 x 9 - 2)."""
    from motif_amd import ops
    x, wt, b = rnd(n, cin, h, w, seed=1), rnd(cin, 2, 4, 4, seed=2, scale=0.1), rnd(2, seed=3)
    close(ops.deconv4x4s2(x.to(dev()), wt.to(dev()), b.to(dev())), F.conv_transpose2d(x, wt, b, 2, 1), 2e-5, 1e-5)


# ------------------------------------------------------------------------------------------- correlations
def test_raft_lookup_vs_alt_corr_restatement_and_corrblock():
    from oracle import native
    from motif_amd.models.core.corr import AlternateCorrBlock, alt_cuda_corr_forward
    B, C, H, W = 2, 128, 16, 24
    f1, f2 = rnd(B, C, H, W, seed=1), rnd(B, C, H, W, seed=2)
    coords = torch.stack(torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")[::-1], 0).float()[None].repeat(B, 1, 1, 1)
    coords = coords + rnd(B, 2, H, W, seed=3, scale=5.0)
    f1h, f2h = f1.permute(0, 2, 3, 1).contiguous(), f2.permute(0, 2, 3, 1).contiguous()
    ch = coords.permute(0, 2, 3, 1).reshape(B, 1, H, W, 2).contiguous()
    ref, = native.alt_corr(f1h, f2h, ch, 3)
    out, = alt_cuda_corr_forward(f1h.to(dev()), f2h.to(dev()), ch.to(dev()), 3)
    close(out, ref, 2e-5, 1e-5, "alt_cuda_corr operator form")
    # full 4-level block vs oracle lookup
    from oracle.motif_ref import alt_corr_lookup
    pyr = [f2]
    for _ in range(3):
        pyr.append(F.avg_pool2d(pyr[-1], 2, stride=2))
    ref4 = alt_corr_lookup(f1, pyr, coords)
    blk = AlternateCorrBlock(f1.to(dev()), f2.to(dev()), radius=3)
    close(blk(coords.to(dev())), ref4, 2e-5, 1e-5, "4-level lookup")


def test_raft_lookup_and_raft_against_reference_corrblock_fixtures():
    """Row C2 with reference-run data only: the HIP pyramid look-up against the output of the reference's own CorrBlock
    (models/core/corr.py:8-56; out-of-range, integer and half-pixel queries), and the HIP RAFT-small against the reference RAFT
    run with alternate_corr=False -- fixtures from tests/golden/make_golden.py:corr_case, no oracle code involved."""
    import argparse
    import numpy as np
    from motif_amd.models.core.corr import AlternateCorrBlock
    from motif_amd.models.core.raft import RAFT
    from motif_amd.utils.synth_weights import synth_tensor
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = dict(np.load(os.path.join(gd, "corrblock_16x24.npz"), allow_pickle=False))
    f1, f2, coords = (torch.from_numpy(g[k]).to(dev()) for k in ("fmap1", "fmap2", "coords"))
    out = AlternateCorrBlock(f1, f2, radius=int(g["radius"]))(coords)
    close(out, torch.from_numpy(g["corr"]), 2e-5, 1e-5, "look-up vs reference CorrBlock")
    g = dict(np.load(os.path.join(gd, "raft_corrblock_128x160.npz"), allow_pickle=False))
    net = RAFT(argparse.Namespace(small=True, mixed_precision=False, alternate_corr=True))
    net.load_state_dict({k: synth_tensor("flow_predictor." + k, v) for k, v in net.state_dict().items()})
    net = net.to(dev()).eval()
    with torch.no_grad():
        lr, up = net(torch.from_numpy(g["image1"]).to(dev()), torch.from_numpy(g["image2"]).to(dev()), iters=int(g["iters"]), test_mode=True)
    close(up, torch.from_numpy(g["flow_up"]), 2e-3, 1e-3, "RAFT vs reference RAFT (CorrBlock path)")
    close(lr, torch.from_numpy(g["flow_lr"]), 5e-4, 1e-3, "RAFT 1/8 flow")


@pytest.mark.parametrize("kernel", ["small", "tiled", "tiled9"])
@pytest.mark.parametrize("shape", [(2, 33, 12, 20), (1, 19, 37, 70), (1, 8, 64, 96), (1, 5, 264, 520)])      # the last: > 512 tiles = the 3-wave form of the tiled kernel
def test_corr81(kernel, shape):
    """Both cost-volume kernels (per-displacement threads for the coarse levels, LDS-tiled + register-blocked for the large
    ones) against the kernel-text restatement: ragged tiles, W % 4 != 0 (scalar edge path), C not a multiple of the chunk."""
    from oracle import native
    from motif_amd import ops
    n, c, h, w = shape
    a, b = rnd(n, c, h, w, seed=1), rnd(n, c, h, w, seed=2)
    ops.set_option("corr81", {"tiled": 1, "small": 2, "tiled9": 3}[kernel])     # the library reads no environment per launch
    try:
        close(ops.corr81(a.to(dev()), b.to(dev())), native.corr81(a, b), 2e-6, 1e-5, "corr81")
        close(ops.corr81(a.to(dev()), b.to(dev()), ops.ACT_LRELU), F.leaky_relu(native.corr81(a, b), 0.1), 2e-6, 1e-5)
    finally:
        ops.set_option("corr81", 0)


# ------------------------------------------------------------------------------------------- SIREN MLPs
def _tables(H, W, HH, WW):
    from motif_amd.models.modules.Ours import gather_tables
    return gather_tables(H, W, HH, WW, dev())


@pytest.mark.parametrize("HW", [((8, 12), (32, 48)), ((7, 9), (14, 18))])
def test_siren_kernels_vs_torch(HW):
    from oracle.motif_ref import Siren as RefSiren
    from motif_amd import ops
    from motif_amd.models.modules.SIREN import Siren
    from motif_amd.utils.synth_weights import fill_state_dict
    (H, W), (HH, WW) = HW
    Q = HH * WW
    iy, ix, rel_y, rel_x = _tables(H, W, HH, WW)
    iyc, ixc = iy.cpu().long(), ix.cpu().long()
    gather = lambda t: t[:, :, iyc][:, :, :, ixc]                                   # [n,c,HH,WW]
    rel = torch.stack([rel_y.cpu()[:, None].expand(HH, WW), rel_x.cpu()[None, :].expand(HH, WW)], 0)   # [2,HH,WW]
    B, N = 2, 2

    class Holder(torch.nn.Module):
        def __init__(self, ref):
            super().__init__()
            self.flow_imnet = RefSiren(67, [64, 64, 256], 3) if ref else Siren(67, [64, 64, 256], 2, 3, True)
            self.imnet = RefSiren(66, [64, 64, 256], 64) if ref else Siren(66, [64, 64, 256], 2, 64, True)
            self.synth_net = RefSiren(198, [64, 64, 64, 256], 3) if ref else Siren(198, [64, 64, 64, 256], 3, 3, True)

    ref, mine = fill_state_dict(Holder(True)), Holder(False)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(dev())
    feat = rnd(2 * B, 64, H, W, seed=1, scale=0.3)
    # imnet
    inp = torch.cat([gather(feat), rel[None].expand(2 * B, -1, -1, -1)], 1)
    with torch.no_grad():
        r = ref.imnet(inp.reshape(2 * B, 66, Q).permute(0, 2, 1)).permute(0, 2, 1).reshape(2 * B, 64, HH, WW)
    o = ops.siren_imnet(mine.imnet.packed(), feat.to(dev()), iy, ix, rel_y, rel_x, HH, WW)
    close(o, r, 5e-6, 1e-4, "imnet")
    l0 = ops.conv2d(mine.imnet.l0_plan(0, 64), feat.to(dev()))          # LR partial of layer 0, then pre=1
    close(ops.siren_imnet(mine.imnet.packed(), l0, iy, ix, rel_y, rel_x, HH, WW, pre=True), r, 5e-6, 1e-4, "imnet pre")
    SPLITS = ((2, "three bf16 parts"), (3, "two fp16 parts"))           # pre = 2 / 3: both forms of the split kernels, same tolerances
    g = rnd(*l0.shape, seed=77).to(dev())
    for pre, what in SPLITS:
        blob = ops.siren_pack_split(ops.SIREN_IMNET, mine.imnet.linears(), pre=pre)
        osp = ops.siren_imnet(blob, l0, iy, ix, rel_y, rel_x, HH, WW, pre=pre)
        close(osp, r, 5e-6, 1e-4, "imnet split, " + what)
        # motif_siren_imnet_add_fwd: + an LR tensor gathered through the same tables, one fp32 add after the head
        h_, w_ = l0.shape[2], l0.shape[3]
        idx = (iy.long()[:, None] * w_ + ix.long()[None, :]).reshape(-1)
        want = osp + g.reshape(l0.shape[0], 64, h_ * w_)[:, :, idx].reshape(l0.shape[0], 64, HH, WW)
        got = ops.siren_imnet(blob, l0, iy, ix, rel_y, rel_x, HH, WW, pre=pre, add_lr=g)
        assert torch.equal(got, want), "imnet + gathered LR term, " + what
    with pytest.raises(RuntimeError):
        ops.siren_imnet(mine.imnet.packed(), l0, iy, ix, rel_y, rel_x, HH, WW, pre=True, add_lr=g)      # split engine only
    # flow_imnet
    times = torch.tensor([[0.0, 0.5], [0.25, 1.0]])
    g = gather(feat).repeat(1, N, 1, 1).reshape(2 * B * N, 64, HH, WW)
    t = times.reshape(B * N, 1, 1, 1).repeat(2, 1, HH, WW)
    inp = torch.cat([g, t, rel[None].expand(2 * B * N, -1, -1, -1)], 1)
    with torch.no_grad():
        r = ref.flow_imnet(inp.reshape(2 * B * N, 67, Q).permute(0, 2, 1)).permute(0, 2, 1).reshape(2 * B * N, 3, HH, WW)
    o = ops.siren_flow(mine.flow_imnet.packed(), feat.to(dev()), iy, ix, rel_y, rel_x, times.to(dev()), N, HH, WW)
    close(o, r, 5e-6, 1e-4, "flow_imnet")
    l0 = ops.conv2d(mine.flow_imnet.l0_plan(0, 64), feat.to(dev()))
    close(ops.siren_flow(mine.flow_imnet.packed(), l0, iy, ix, rel_y, rel_x, times.to(dev()), N, HH, WW, pre=True), r, 5e-6, 1e-4, "flow pre")
    for pre, what in SPLITS:
        blob = ops.siren_pack_split(ops.SIREN_FLOW, mine.flow_imnet.linears(), pre=pre)
        close(ops.siren_flow(blob, l0, iy, ix, rel_y, rel_x, times.to(dev()), N, HH, WW, pre=pre), r, 5e-6, 1e-4, "flow split, " + what)
    # synth (with the normalisation prologue): build an accumulator with zeros / ones / exact-equality cases
    acc = rnd(B * N, 133, HH, WW, seed=5, scale=0.5)
    acc[:, 130] = acc[:, 130].abs() * 2 + 1e-3
    acc[:, :130] *= acc[:, 130:131]          # sums scale with the normaliser, as real splat sums do
    acc[:, 130, :2] = 0.0
    acc[:, :130, :2] = 0.0
    acc[:, 130, 2:4] = 1.0
    acc[:, 131] = 1.0 + acc[:, 131].abs()
    acc[:, 132] = torch.randint(0, 9, (B * N, HH, WW), generator=torch.Generator().manual_seed(6)).float()
    res = rnd(B, 64, H, W, seed=7, scale=0.3)
    wz = acc[:, 130:131].clone()
    wz[wz == 0] = 1.0
    out = acc[:, :130] / wz
    cnt = acc[:, 132:133]
    cnt_ = cnt.clone()
    cnt_[cnt_ == 0.0] = 1.0
    wz_ = wz.clone()
    wz_[wz_ == 1.0] = 0.0
    extra = torch.cat((acc[:, 131:132], cnt / 16.0, wz_ / cnt_), 1)
    allin = torch.cat((out, extra, gather(res).repeat(1, N, 1, 1).reshape(B * N, 64, HH, WW), times.reshape(B * N, 1, 1, 1).repeat(1, 1, HH, WW)), 1)
    si = ops.synth_input(acc.to(dev()), res.to(dev()), iy, ix, times.to(dev()), B, N, HH, WW)
    close(si, allin, 1e-6, 1e-6, "synth input / normalisation")
    with torch.no_grad():
        r = ref.synth_net(allin.reshape(B * N, 198, Q).permute(0, 2, 1)).permute(0, 2, 1).reshape(B, N, 3, HH, WW).permute(1, 0, 2, 3, 4).clamp(0, 1)
    o = ops.siren_synth(mine.synth_net.packed(), acc.to(dev()), res.to(dev()), iy, ix, times.to(dev()), B, N, HH, WW)
    close(o, r, 2e-5, 1e-4, "synth")
    l0 = ops.conv2d(mine.synth_net.l0_plan(133, 197), res.to(dev()))
    close(ops.siren_synth(mine.synth_net.packed(), acc.to(dev()), l0, iy, ix, times.to(dev()), B, N, HH, WW, pre=True), r, 2e-5, 1e-4, "synth pre")
    for pre, what in SPLITS:
        blob = ops.siren_pack_split(ops.SIREN_SYNTH, mine.synth_net.linears(), pre=pre)
        close(ops.siren_synth(blob, acc.to(dev()), l0, iy, ix, times.to(dev()), B, N, HH, WW, pre=pre), r, 2e-5, 1e-4, "synth split, " + what)


def test_siren_kernels_are_reproducible_from_run_to_run():
    """The three MLP kernels at the c2 size, six launches each on the same inputs: bit-identical outputs, in the default two-part form and
    in the three-part form.  (Round 5: the two-part flow kernel lost this property when its sine lost the v_fract in front -- a third of
    the pixels differed by up to 9e-6 from launch to launch, cause not found; siren_split.hip keeps the fract and this test pins it.)"""
    from motif_amd import ops
    from motif_amd.models.modules.SIREN import Siren
    from motif_amd.utils.synth_weights import fill_state_dict
    H, W, HH, WW = 180, 320, 720, 1280
    iy, ix, rel_y, rel_x = _tables(H, W, HH, WW)

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.flow_imnet = Siren(67, [64, 64, 256], 2, 3, True)
            self.imnet = Siren(66, [64, 64, 256], 2, 64, True)
            self.synth_net = Siren(198, [64, 64, 64, 256], 3, 3, True)
    mine = fill_state_dict(Holder()).to(dev())
    B, N = 1, 3
    feat = rnd(2 * B, 64, H, W, seed=1, scale=0.3).to(dev())
    times = torch.tensor([[0.25, 0.5, 1.0]]).to(dev())
    acc = rnd(B * N, 133, HH, WW, seed=5, scale=0.5)
    acc[:, 130] = acc[:, 130].abs() * 2 + 1e-3
    acc[:, :130] *= acc[:, 130:131]
    acc[:, 131] = 1.0 + acc[:, 131].abs()
    acc[:, 132] = torch.randint(0, 9, (B * N, HH, WW), generator=torch.Generator().manual_seed(6)).float()
    acc = acc.to(dev())
    res = rnd(B, 64, H, W, seed=7, scale=0.3).to(dev())
    for pre in (3, 2):
        fl0 = ops.conv2d(mine.flow_imnet.l0_plan(0, 64), feat)
        il0 = ops.conv2d(mine.imnet.l0_plan(0, 64), feat)
        sl0 = ops.conv2d(mine.synth_net.l0_plan(133, 197), res)
        fb, ib, sb = (ops.siren_pack_split(k, m.linears(), pre=pre) for k, m in ((ops.SIREN_FLOW, mine.flow_imnet), (ops.SIREN_IMNET, mine.imnet), (ops.SIREN_SYNTH, mine.synth_net)))
        runs = {"flow_imnet": lambda: ops.siren_flow(fb, fl0, iy, ix, rel_y, rel_x, times, N, HH, WW, pre=pre),
                "imnet": lambda: ops.siren_imnet(ib, il0, iy, ix, rel_y, rel_x, HH, WW, pre=pre),
                "synth_net": lambda: ops.siren_synth(sb, acc, sl0, iy, ix, times, B, N, HH, WW, pre=pre)}
        for name, fn in runs.items():
            ref = fn().clone()
            for _ in range(5):
                assert torch.equal(fn(), ref), "%s (pre = %d) differs from launch to launch" % (name, pre)


# ------------------------------------------------------------------------------------------- fused MoTIF splat
@pytest.mark.parametrize("far", [False, True])
def test_splat_motif_owner_computes_vs_kernel_text(far):
    """motif_splat_motif_fwd (owner-computes tiles + far-source fallback) == the reference composition
    cat(feat*e^z, e^z) -> sum splat, max splat, count splat, both directions added (Ours.py:777-816)."""
    from oracle import native
    from motif_amd import ops
    B, N, H, W, s = 2, 2, 12, 20, 4
    HH, WW = H * s, W * s
    Q = HH * WW
    iy, ix, _, _ = _tables(H, W, HH, WW)
    iyc, ixc = iy.cpu().long(), ix.cpu().long()
    imnet_out = rnd(2 * B, 64, HH, WW, seed=1)
    feat_lr = rnd(2 * B, 64, H, W, seed=2)
    pred = rnd(2 * B * N, 3, HH, WW, seed=3, scale=0.05)
    if far:
        pred[:, :2, 5:9, 7:30] *= 12.0       # |flow| up to ~48 px > the 16 px owner halo -> fallback path
        pred[0, 0, 20, 40] = 1e6              # absurd flow: must simply vanish
    alpha = torch.tensor([-20.0])
    flow = pred[:, :2] * 20.0 * (HH / H)
    z = F.relu(pred[:, 2:3]) * alpha
    ez = z.exp()
    feat_low = feat_lr[:, :, iyc][:, :, :, ixc]
    rep = lambda t: t.repeat(1, N, 1, 1).reshape(2 * B * N, -1, HH, WW)
    feat_all = torch.cat([rep(imnet_out), pred[:, :2], rep(feat_low)], 1)
    ssum = native.splat(torch.cat([feat_all * ez, ez], 1), flow, "sum").reshape(2, B * N, 131, HH, WW).sum(0)
    smax = native.splat(ez, flow, "max").reshape(2, B * N, 1, HH, WW).max(0)[0]
    scnt = native.splat(torch.ones_like(ez), flow, "count").reshape(2, B * N, 1, HH, WW).sum(0)
    acc = ops.splat_motif(imnet_out.to(dev()), pred.to(dev()), feat_lr.to(dev()), iy, ix, alpha.to(dev()), HH / H, B, N, HH, WW).cpu()
    assert torch.equal(acc[:, 132:133], scnt), "count plane must be exact"
    close(acc[:, 131:132], smax, 1e-6, 1e-6, "max plane")
    close(acc[:, :131], ssum, 3e-5, 1e-5, "sum planes")


def test_splat_motif_keeps_dynamic_range_of_low_reliability_sources():
    """Occluded / unreliable sources: relu(pred2) in [0, 3] with alpha = -20 gives e^z down to e^-60 (Ours.py:794).
    The reference's fp32 atomics keep such addends (softsplat_cp.py:35-50) and the post-splat normalisation
    (Ours.py:811-830) divides them out again, so cells reached ONLY by unreliable sources still get a correctly
    normalised feature.  Bars: hit count exact; warped_z == 0 set identical; warped_z and z_max within 3e-5 relative
    (no absolute floor: the values span 26 decades); normalised output within 3e-5; two runs bit-identical."""
    from oracle import native
    from motif_amd import ops
    B, N, H, W, s = 1, 2, 16, 24, 4
    HH, WW = H * s, W * s
    iy, ix, _, _ = _tables(H, W, HH, WW)
    iyc, ixc = iy.cpu().long(), ix.cpu().long()
    imnet_out = rnd(2 * B, 64, HH, WW, seed=1)
    feat_lr = rnd(2 * B, 64, H, W, seed=2)
    pred = rnd(2 * B * N, 3, HH, WW, seed=3, scale=0.05)
    p2 = torch.rand(2 * B * N, HH, WW, generator=torch.Generator().manual_seed(4)) * 3.0          # z in [-60, 0]
    p2[:, 10:30, 20:60] = 2.0 + p2[:, 10:30, 20:60] / 3.0      # a region where EVERY contributor is tiny (e^z < 4e-18)
    p2[:, 40:44] = 0.0                                         # and fully reliable rows next to it
    p2[0, 50:54, 30:40] = -1.0                                 # relu clamps negatives: e^z = 1
    pred[:, 2] = p2
    pred[0, 0, 6, 10], pred[0, 1, 6, 10] = 0.3, 0.2            # one far source (24, 16 px) -> fp32 global-atomic fallback path
    alpha = torch.tensor([-20.0])
    flow = pred[:, :2] * 20.0 * (HH / H)
    ez = (F.relu(pred[:, 2:3]) * alpha).exp()
    assert float(ez.min()) < 1e-25
    feat_low = feat_lr[:, :, iyc][:, :, :, ixc]
    rep = lambda t: t.repeat(1, N, 1, 1).reshape(2 * B * N, -1, HH, WW)
    feat_all = torch.cat([rep(imnet_out), pred[:, :2], rep(feat_low)], 1)
    ssum = native.splat(torch.cat([feat_all * ez, ez], 1), flow, "sum").reshape(2, B * N, 131, HH, WW).sum(0)
    smax = native.splat(ez, flow, "max").reshape(2, B * N, 1, HH, WW).max(0)[0]
    scnt = native.splat(torch.ones_like(ez), flow, "count").reshape(2, B * N, 1, HH, WW).sum(0)
    args = (imnet_out.to(dev()), pred.to(dev()), feat_lr.to(dev()), iy, ix, alpha.to(dev()), HH / H, B, N, HH, WW)
    acc = ops.splat_motif(*args).cpu()
    assert torch.equal(acc, ops.splat_motif(*args).cpu()), "owner-computes splat must be run-to-run bit-identical"
    assert torch.equal(acc[:, 132:133], scnt), "count plane must be exact"
    wz_ref, wz = ssum[:, 130:131], acc[:, 130:131]
    assert torch.equal(wz == 0, wz_ref == 0), "warped_z == 0 set (Ours.py:811) differs"
    hit_only_tiny = (wz_ref > 0) & (wz_ref < 1e-12)
    assert int(hit_only_tiny.sum()) > 500, "test must reach cells whose every contributor is unreliable"
    rel = lambda x, r: float(((x - r).abs() / r.abs().clamp_min(1e-37)).max())
    assert rel(wz, wz_ref) < 3e-5, rel(wz, wz_ref)
    assert rel(acc[:, 131:132], smax) < 3e-5
    assert torch.equal(acc[:, 131:132] == 1.0, smax == 1.0)
    one = lambda t: torch.where(t == 0, torch.ones_like(t), t)
    out_ref, out = ssum[:, :130] / one(wz_ref), acc[:, :130] / one(wz)          # Ours.py:811-814
    close(out, out_ref, 3e-5, 3e-5, "normalised splat output")
    close(out[hit_only_tiny.expand_as(out)], out_ref[hit_only_tiny.expand_as(out)], 3e-5, 3e-5, "normalised output, unreliable-only cells")


def _flow_values_where_fma_and_rounded_product_disagree(sr, x, n=400000, seed=0):
    """pred values p for which floor(x + fl(fl(20 p) sr)) != floor(fma(fl(20 p), sr, x)): the reference multiplies its flow
    tensor out first (Ours.py:794) and the kernel adds the index afterwards (softsplat_cp.py:27-28)."""
    rng = np.random.default_rng(seed)
    k = rng.integers(1, 6, n)
    p = ((k + rng.uniform(-1.2e-6, 1.2e-6, n)) / (20.0 * sr)).astype(np.float32)
    q = (p * np.float32(20.0)).astype(np.float32)
    sep = (np.float32(x) + (q * np.float32(sr)).astype(np.float32)).astype(np.float32)
    fused = (np.float64(x) + q.astype(np.float64) * np.float64(np.float32(sr))).astype(np.float32)
    return np.unique(p[np.floor(sep) != np.floor(fused)])


def test_splat_motif_rounds_the_flow_before_adding_the_pixel_index():
    """Scale ratio 3 (not a power of two): the flow product must be rounded to fp32 BEFORE the pixel index is added, as
    the reference's separate torch multiplies do -- a fused multiply-add moves some targets across an integer, i.e. to
    other cells.  Sources crafted to sit on such boundaries; the count plane (one per footprint corner) must be exact."""
    from oracle import native
    from motif_amd import ops
    B, N, H, W, s = 1, 1, 16, 24, 3
    HH, WW = H * s, W * s
    cases = [(x, v) for x in range(2, 40) for v in _flow_values_where_fma_and_rounded_product_disagree(float(s), x, n=100000)]
    assert len(cases) >= 40, "search found too few boundary cases"
    iy, ix, _, _ = _tables(H, W, HH, WW)
    imnet_out = rnd(2 * B, 64, HH, WW, seed=1)
    feat_lr = rnd(2 * B, 64, H, W, seed=2)
    pred = rnd(2 * B * N, 3, HH, WW, seed=3, scale=0.02)
    for i, (x, v) in enumerate(cases):
        pred[0, 0, 2 + i % 40, x] = float(v)
        pred[0, 1, 2 + i % 40, x] = 0.0
    alpha = torch.tensor([-20.0])
    flow = pred[:, :2] * 20.0 * (HH / H)
    ez = (F.relu(pred[:, 2:3]) * alpha).exp()
    scnt = native.splat(torch.ones_like(ez), flow, "count").reshape(2, B * N, 1, HH, WW).sum(0)
    acc = ops.splat_motif(imnet_out.to(dev()), pred.to(dev()), feat_lr.to(dev()), iy, ix, alpha.to(dev()), HH / H, B, N, HH, WW).cpu()
    assert torch.equal(acc[:, 132:133], scnt), "count plane differs: flow not rounded before the index add"


def test_splat_motif_sinks_and_overfull_tiles():
    """The owner-computes tiles against the kernel text on flows that stress their bookkeeping: a point sink (hundreds of
    footprints on one cell), every source within 14 px of a tile landing inside it (7.7 k sources for 1 k cells), motion
    boundaries, a far source.  Count exact, max and sums within 3e-5, two runs bit-identical."""
    from oracle import native
    from motif_amd import ops
    B, N, H, W, s = 1, 1, 32, 48, 4
    HH, WW = H * s, W * s
    iy, ix, _, _ = _tables(H, W, HH, WW)
    iyc, ixc = iy.cpu().long(), ix.cpu().long()
    imnet_out = rnd(2 * B, 64, HH, WW, seed=1)
    feat_lr = rnd(2 * B, 64, H, W, seed=2)
    alpha = torch.tensor([-20.0])

    def run(pred):
        return ops.splat_motif(imnet_out.to(dev()), pred.to(dev()), feat_lr.to(dev()), iy, ix, alpha.to(dev()), HH / H, B, N, HH, WW).cpu()

    def check(pred, reproducible):
        flow = pred[:, :2] * 20.0 * (HH / H)
        ez = (F.relu(pred[:, 2:3]) * alpha).exp()
        feat_low = feat_lr[:, :, iyc][:, :, :, ixc]
        feat_all = torch.cat([imnet_out, pred[:, :2], feat_low], 1)
        ssum = native.splat(torch.cat([feat_all * ez, ez], 1), flow, "sum").reshape(2, B * N, 131, HH, WW).sum(0)
        smax = native.splat(ez, flow, "max").reshape(2, B * N, 1, HH, WW).max(0)[0]
        scnt = native.splat(torch.ones_like(ez), flow, "count").reshape(2, B * N, 1, HH, WW).sum(0)
        acc = run(pred)
        assert torch.equal(acc[:, 132:133], scnt), "count plane must be exact"
        close(acc[:, 131:132], smax, 1e-6, 1e-6, "max plane")
        scale = float(ssum.abs().max())
        close(acc[:, :131], ssum, 3e-5 * max(1.0, scale / 50), 3e-5, "sum planes")
        if reproducible:
            assert torch.equal(acc, run(pred)), "owner-computes tiles must be run-to-run bit-identical"
        return scnt

    yy, xx = torch.meshgrid(torch.arange(HH, dtype=torch.float32), torch.arange(WW, dtype=torch.float32), indexing="ij")
    k = 1.0 / (20.0 * s)                                                   # pred units per HR pixel of flow
    # (1) point sink: a 24 x 24 block of both directions flows to (40.3, 70.6)
    pred = rnd(2, 3, HH, WW, seed=3, scale=0.02)
    blk = (slice(28, 52), slice(58, 82))
    pred[:, 0][(slice(None),) + blk] = ((70.6 - xx) * k)[blk]
    pred[:, 1][(slice(None),) + blk] = ((40.3 - yy) * k)[blk]
    pred[:, 2] = torch.rand(2, HH, WW, generator=torch.Generator().manual_seed(4)) * 0.2
    cnt = check(pred, reproducible=True)
    assert float(cnt.max()) > 500, "the sink must pile hundreds of footprints on one cell"
    # (2) everything within 14 px of the tile rows 48..63 x columns 64..127 lands inside it; a far source on top (its
    #     global float atomics are the one order-dependent part)
    pred = rnd(2, 3, HH, WW, seed=5, scale=0.02)
    blk = (slice(35, 77), slice(51, 141))
    pred[:, 0][(slice(None),) + blk] = ((xx.clamp(65.5, 125.5) - xx) * k)[blk]
    pred[:, 1][(slice(None),) + blk] = ((yy.clamp(49.5, 61.5) - yy) * k)[blk]
    pred[:, 2] = 0.0
    pred[0, 0, 100, 20] = 0.5
    cnt = check(pred, reproducible=True)
    assert float(cnt[..., 48:64, 64:128].sum()) > 4 * 7000


def test_splat_motif_value_clamp_and_non_finite_inputs():
    """Defined behaviour of the fused splat at the edges of its fixed-point accumulation (splat.hip, include/motif_hip.h):
      * plane values are clamped to +-2^17 when they are staged (the exact integer sums need |addend * 2^(32-E)| < 2^51): a source
        value of 1e6 contributes as 131072, +-Inf as +-131072; the result equals the kernel-text splat of the CLAMPED sources;
      * a NaN plane value does not propagate either -- the accumulator stays finite (the reference's float atomicAdd would leave a
        NaN in the touched cells; its driver never produces one: the sources are SIREN outputs);
      * a source whose flow is not finite is DROPPED: it touches no cell, the hit count included (the reference asserts finite
        flows up front, softsplat_cp.py:25-26; a kernel cannot raise, and floor(NaN) names no cell)."""
    from oracle import native
    from motif_amd import ops
    B, N, H, W, s = 1, 1, 16, 24, 4
    HH, WW = H * s, W * s
    iy, ix, _, _ = _tables(H, W, HH, WW)
    iyc, ixc = iy.cpu().long(), ix.cpu().long()
    imnet_out = rnd(2 * B, 64, HH, WW, seed=1)
    feat_lr = rnd(2 * B, 64, H, W, seed=2)
    alpha = torch.tensor([-20.0])
    pred = rnd(2, 3, HH, WW, seed=3, scale=0.01)
    pred[:, 2] = 0.0
    M = 131072.0

    def run(im, pr):
        return ops.splat_motif(im.to(dev()), pr.to(dev()), feat_lr.to(dev()), iy, ix, alpha.to(dev()), HH / H, B, N, HH, WW).cpu()

    def ref(im, pr):
        flow = pr[:, :2] * 20.0 * (HH / H)
        ez = (F.relu(pr[:, 2:3]) * alpha).exp()
        feat_all = torch.cat([im, pr[:, :2], feat_lr[:, :, iyc][:, :, :, ixc]], 1)
        ssum = native.splat(torch.cat([feat_all * ez, ez], 1), flow, "sum").reshape(2, B * N, 131, HH, WW).sum(0)
        scnt = native.splat(torch.ones_like(ez), flow, "count").reshape(2, B * N, 1, HH, WW).sum(0)
        return ssum, scnt

    def rel(a, b):
        return float(((a - b).abs() / (1 + b.abs())).max())

    im = imnet_out.clone()
    im[0, 5, 20, 30], im[1, 7, 40, 50], im[0, 9, 10, 10], im[0, 11, 12, 12] = 1e6, -3e5, M, 200000.0
    im[1, 3, 30, 40], im[0, 2, 44, 70] = float("inf"), float("-inf")
    acc = run(im, pred)
    ssum, scnt = ref(im.clamp(-M, M), pred)
    assert torch.isfinite(acc).all()
    assert torch.equal(acc[:, 132:133], scnt)
    assert rel(acc[:, :131], ssum) < 1e-6, "values beyond +-2^17 must contribute as +-2^17"
    assert float((acc[:, :131] - ref(im.nan_to_num(posinf=3e38, neginf=-3e38), pred)[0]).abs().max()) > 1e5       # i.e. NOT the unclamped sum
    im2 = imnet_out.clone()
    im2[0, 5, 20, 30] = float("nan")
    assert torch.isfinite(run(im2, pred)).all()
    ssum0, scnt0 = ref(imnet_out, pred)
    for bad in (float("nan"), float("inf"), float("-inf")):
        pr2 = pred.clone()
        pr2[0, 0, 20, 30], pr2[1, 1, 33, 44] = bad, bad
        a3 = run(imnet_out, pr2)
        pr3 = pred.clone()
        pr3[0, 0, 20, 30], pr3[1, 1, 33, 44] = 1e3, 1e3                     # the same two sources sent far outside: no contribution
        ssum3, scnt3 = ref(imnet_out, pr3)
        assert torch.isfinite(a3).all()
        assert torch.equal(a3[:, 132:133], scnt3) and not torch.equal(scnt3, scnt0)
        assert rel(a3[:, :131], ssum3) < 1e-6


def test_precontracted_splat_and_synth_equal_the_literal_path():
    """motif_splat_motif_pre_fwd + motif_siren_synth_pre_fwd (synth_net's first layer contracted into the splat sources:
    Ours.py:811-814 and 839-856 are linear) against the literal composition on the CPU: kernel-text splat of the 130
    planes -> normalise -> the oracle Siren.  Includes unreliable-only cells, far sources and empty cells; hit count and
    max planes exact, frames within 2e-5 + 1e-4 rel, two runs bit-identical."""
    from oracle import native
    from oracle.motif_ref import Siren as RefSiren
    from motif_amd import ops
    from motif_amd.models.modules.SIREN import Siren
    from motif_amd.utils.synth_weights import fill_state_dict
    B, N, H, W, s = 1, 2, 12, 20, 4
    HH, WW = H * s, W * s
    Q = HH * WW
    iy, ix, _, _ = _tables(H, W, HH, WW)
    iyc, ixc = iy.cpu().long(), ix.cpu().long()
    gather = lambda t: t[:, :, iyc][:, :, :, ixc]

    class Holder(torch.nn.Module):
        def __init__(self, ref):
            super().__init__()
            self.synth_net = RefSiren(198, [64, 64, 64, 256], 3) if ref else Siren(198, [64, 64, 64, 256], 3, 3, True)

    ref, mine = fill_state_dict(Holder(True)), Holder(False)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(dev())
    imnet_out = rnd(2 * B, 64, HH, WW, seed=1)
    feat_lr = rnd(2 * B, 64, H, W, seed=2)
    res = rnd(B, 64, H, W, seed=7, scale=0.3)
    pred = rnd(2 * B * N, 3, HH, WW, seed=3, scale=0.05)
    p2 = torch.rand(2 * B * N, HH, WW, generator=torch.Generator().manual_seed(4)) * 0.2
    p2[:, 10:20, 20:50] = 2.0                                   # cells reached by unreliable sources only
    pred[:, 2] = p2
    pred[:, :2, 24:40, 6:50] = 0.0
    pred[:, 1, 24:40, 6:50] = 0.11                              # a block that moves 8.8 px down: leaves empty cells behind
    pred[0, 0, 6, 10], pred[0, 1, 6, 10] = 0.3, 0.2             # one far source
    alpha = torch.tensor([-20.0])
    times = torch.tensor([[0.25, 0.75]])
    flow = pred[:, :2] * 20.0 * (HH / H)
    ez = (F.relu(pred[:, 2:3]) * alpha).exp()
    rep = lambda t: t.repeat(1, N, 1, 1).reshape(2 * B * N, -1, HH, WW)
    feat_all = torch.cat([rep(imnet_out), pred[:, :2], rep(gather(feat_lr))], 1)
    ssum = native.splat(torch.cat([feat_all * ez, ez], 1), flow, "sum").reshape(2, B * N, 131, HH, WW).sum(0)
    smax = native.splat(ez, flow, "max").reshape(2, B * N, 1, HH, WW).max(0)[0]
    scnt = native.splat(torch.ones_like(ez), flow, "count").reshape(2, B * N, 1, HH, WW).sum(0)
    assert int((scnt == 0).sum()) > 50
    wz = ssum[:, 130:131].clone()
    wz[wz == 0] = 1.0
    cnt_ = scnt.clone()
    cnt_[cnt_ == 0] = 1.0
    wz_ = wz.clone()
    wz_[wz_ == 1.0] = 0.0
    allin = torch.cat((ssum[:, :130] / wz, smax, scnt / 16.0, wz_ / cnt_, gather(res).repeat(1, N, 1, 1).reshape(B * N, 64, HH, WW),
                       times.reshape(B * N, 1, 1, 1).repeat(1, 1, HH, WW)), 1)
    with torch.no_grad():
        r = ref.synth_net(allin.reshape(B * N, 198, Q).permute(0, 2, 1)).permute(0, 2, 1).reshape(B, N, 3, HH, WW).permute(1, 0, 2, 3, 4).clamp(0, 1)
    # the device side: W0 split as LunaTokis._pre_plan does it
    w0 = mine.synth_net.net[0].linear.weight.detach()
    u_hr = torch.einsum("ck,bkhw->bchw", w0[:, :64].double().cpu(), imnet_out.double()).float()       # stands for the composed imnet head
    g_lr = ops.conv2d(ops.ConvPlan(w0[:, 66:130].contiguous().view(64, 64, 1, 1), None), feat_lr.to(dev()))
    ab = torch.stack([w0[:, 64], w0[:, 65]]).contiguous()
    args = (u_hr.to(dev()), pred.to(dev()), g_lr, ab, iy, ix, alpha.to(dev()), HH / H, B, N, HH, WW)
    acc = ops.splat_motif_pre(*args)
    assert acc.shape == (B * N, 67, HH, WW)
    assert torch.equal(acc, ops.splat_motif_pre(*args)), "run-to-run bit identity"
    assert torch.equal(acc[:, 66:67].cpu(), scnt), "count plane must be exact"
    close(acc[:, 65:66], smax, 0, 3e-5, "max plane")
    relz = ((acc[:, 64:65].cpu() - ssum[:, 130:131]).abs() / ssum[:, 130:131].abs().clamp_min(1e-37)).max()
    assert float(relz) < 3e-5 and torch.equal(acc[:, 64:65].cpu() == 0, ssum[:, 130:131] == 0)
    pre_ref = torch.einsum("ck,bkhw->bchw", w0[:, :130].double().cpu(), (ssum[:, :130] / wz).double()).float()
    close(acc[:, :64].cpu() / wz, pre_ref, 3e-5, 3e-5, "contracted first-layer sums")
    l0 = ops.conv2d(mine.synth_net.l0_plan(133, 197), res.to(dev()))
    for pre in (2, 3):                                                   # three bf16 parts / two fp16 parts
        blob = ops.siren_pack_split(ops.SIREN_SYNTH_PRE, mine.synth_net.linears(), pre=pre)
        o = ops.siren_synth_pre(blob, acc, l0, iy, ix, times.to(dev()), B, N, HH, WW, pre=pre)
        close(o, r, 2e-5, 1e-4, "synth on the pre-contracted accumulator (pre = %d)" % pre)
    # fused-source form (what LunaTokis runs): U already holds U + G -- the same fp32 sum the kernel forms from g_lr -- and g_lr = None
    up = iy.long()[:, None] * W + ix.long()[None, :]
    ug = u_hr.to(dev()) + g_lr.reshape(2 * B, 64, H * W)[:, :, up.reshape(-1)].reshape(2 * B, 64, HH, WW)
    accf = ops.splat_motif_pre(ug, pred.to(dev()), None, ab, iy, ix, alpha.to(dev()), HH / H, B, N, HH, WW, lr_size=(H, W))
    assert torch.equal(accf, acc), "U + G added upstream must give the accumulator of the g_lr form bit for bit"
    # accumulate form: the same two directions added twice == sums doubled, count doubled, max unchanged
    acc2 = ops.splat_motif_pre(*args, acc=acc.clone(), accumulate=True)
    assert torch.equal(acc2[:, 66], 2 * acc[:, 66]) and torch.equal(acc2[:, 65], acc[:, 65])
    close(acc2[:, :65], 2 * acc[:, :65], 1e-30, 1e-6, "accumulate")


def test_dcn_fused_multi_vs_kernel_text(engine):
    """Fused DCN (deformable im2col in LDS + MFMA, multi-problem) == the kernel-text restatement, including
    offsets that leave the image, the (-1, 0) border band, and odd image sizes.  engine = bf16x3 runs
    dcn_fused_kernel<8,true> (the bench default: GEMM on the bf16 matrix cores), fp32 runs dcn_fused_kernel<8,false>."""
    from oracle import native
    from motif_amd import ops
    assert ops.get_conv_mma() == {"bf16x3": ops.MMA_BF16X3, "f16x2": ops.MMA_F16X2, "fp32": ops.MMA_FP32}[engine]      # f16x2: the DCN GEMM stays three-part
    assert not os.environ.get("MOTIF_DCN_UNFUSED") and not os.environ.get("MOTIF_DCN_FP32")
    B, C, H, W, dg, P = 2, 64, 21, 45, 8, 3
    outs_ref, plans, xs, oms = [], [], [], []
    for pi in range(P):
        x = rnd(B, C, H, W, seed=10 * pi + 1)
        w = rnd(64, C, 3, 3, seed=10 * pi + 2, scale=0.05)
        bias = rnd(64, seed=10 * pi + 3, scale=0.1)
        off = rnd(B, 2 * dg * 9, H, W, seed=10 * pi + 4, scale=4.0)
        off[:, :, :3] *= 8.0                                   # far outside the image
        off[:, ::2, 5, :] = -0.5 - torch.arange(W) * 0.0       # sample rows in the (-1, 0) band for row 5 taps
        mlog = rnd(B, dg * 9, H, W, seed=10 * pi + 5, scale=2.0)
        outs_ref.append(F.leaky_relu(native.dcn_v2_forward(x, w, bias, off, torch.sigmoid(mlog), 3, 3, 1, 1, 1, 1, 1, 1, dg), 0.1))
        plans.append(ops.DcnPlan(w.to(dev()), bias.to(dev())))
        xs.append(x.to(dev()))
        oms.append(torch.cat([off, torch.sigmoid(mlog)], 1).to(dev()))
    out = ops.dcn_v2_multi(plans, xs, oms, dg, ops.ACT_LRELU)
    for pi in range(P):
        close(out[pi], outs_ref[pi], 3e-5, 3e-5, "fused dcn problem %d" % pi)


@pytest.mark.parametrize("form", ["window8", "fused4_bf16x3", "fused4_fp32", "fused8_oddW"])
def test_dcn_concurrent_with_conv_split_is_bit_identical(form, keep_mma):
    """The fused DCN launched on one stream while the 3x3 conv kernels of the shapes RAFT / the trunk issue run on another:
    every one of 4 x 100 outputs must equal the serial result bit for bit.  Regression guard for the co-residency corruption
    reported in round 1 (dcn.hip, launch comment).  Each form is FORCED through the library options so that the guard really
    runs the kernel it names: the shipped window kernel (dcn_win_kernel<8>), the 4-wave fused form the report was about
    (dcn_fused_kernel<4,*>: two blocks per CU) in both engines, and the 8-wave fallback on a map whose width is not a multiple
    of 4 (what a PCD level of width 162 takes)."""
    from motif_amd import ops
    from motif_amd.models.modules.DCNv2.dcn_v2 import DCN_sep
    from motif_amd.models.modules.layers import Conv2d
    ops.set_mma("fp32" if form == "fused4_fp32" else "bf16x3")
    torch.manual_seed(0)
    gru = Conv2d(242, 96, 3, 1, 1).to(dev())
    xg = torch.randn(2, 242, 90, 160, device=dev())
    tr = Conv2d(64, 64, 3, 1, 1).to(dev())
    xt = torch.randn(4, 64, 360, 640, device=dev())
    dcns = [DCN_sep(64, 64, 3, stride=1, padding=1, dilation=1, deformable_groups=8).to(dev()) for _ in range(2)]
    with torch.no_grad():
        for d in dcns:
            d.conv_offset_mask.weight.normal_(0, 0.05)
    h, w = (180, 322) if form == "fused8_oddW" else (180, 320)
    xs = [torch.randn(1, 64, h, w, device=dev()) for _ in range(2)]
    feas = [torch.randn(1, 64, h, w, device=dev()) for _ in range(2)]
    try:
        if form.startswith("fused4"):
            ops.set_option("dcn_nowin", 1)
            ops.set_option("dcn_waves", 4)
        with torch.no_grad():
            oms = ops.conv2d_multi([d.conv_offset_mask.plan() for d in dcns], feas, act=ops.ACT_NONE, act2=ops.ACT_SIGMOID, act_split=144)
            oms = [oms[0].clone(), oms[1].clone()]
            ref = ops.dcn_v2_multi([d.dplan() for d in dcns], xs, oms, 8, ops.ACT_LRELU).clone()
            torch.cuda.synchronize()
            sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
            bad = 0
            for _ in range(100):
                with torch.cuda.stream(sb):
                    gru(xg, act=1)
                    tr(xt, act=1)
                with torch.cuda.stream(sa):
                    outs = [ops.dcn_v2_multi([d.dplan() for d in dcns], xs, oms, 8, ops.ACT_LRELU) for _ in range(4)]
                torch.cuda.synchronize()
                bad += sum(int(not torch.equal(o, ref)) for o in outs)
    finally:
        ops.set_option("dcn_nowin", 0)
        ops.set_option("dcn_waves", 0)
    assert bad == 0, "%d of 400 concurrent DCN outputs (%s) differ from the serial result" % (bad, form)


# ------------------------------------------------------------------------------------------- frame formats
def test_frame_decode_encode_bit_exact():
    """motif_frames_u8_to_f32 / _f32_to_u8 against the numpy restatement of the reference's loader and tensor2img /
    demo.py encoders: bit-exact, including the .5 rounding ties, out-of-range values and every uint8 level."""
    from oracle import frames_ref
    from motif_amd import ops
    rng = np.random.default_rng(1)
    u8 = rng.integers(0, 256, (3, 37, 53, 3), dtype=np.uint8)
    u8[0, 0, :256 // 5 + 1] = np.arange(0, 256, 5, dtype=np.uint8)[:, None][: 256 // 5 + 1]
    dec = ops.frames_u8_to_f32(torch.from_numpy(u8).to(dev()))
    assert torch.equal(dec.cpu(), torch.from_numpy(frames_ref.decode(u8)))
    x = rng.random((3, 3, 37, 53), dtype=np.float32) * 1.4 - 0.2
    x[0, :, 0, :40] = (np.arange(40, dtype=np.float32) + 0.5) / 255.0          # exact ties
    xt = torch.from_numpy(x).to(dev())
    assert np.array_equal(ops.frames_f32_to_u8(xt).cpu().numpy(), frames_ref.encode_tensor2img(x))
    assert np.array_equal(ops.frames_f32_to_u8(xt, round_half_even=False, swap_rb=False).cpu().numpy(), frames_ref.encode_demo(x))
    # codec round trip: every level survives decode -> encode
    lv = torch.arange(256, dtype=torch.uint8).view(1, 1, 256, 1).repeat(1, 2, 1, 3).to(dev())
    assert torch.equal(ops.frames_f32_to_u8(ops.frames_u8_to_f32(lv)), lv)
