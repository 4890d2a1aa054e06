"""Tensor-level wrappers over the C ABI (include/motif_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every computation below is a
hand-written HIP kernel in libmotif_hip.so.  All tensors must be CUDA (ROCm) fp32; there is no CPU or
eager fallback -- a missing library or a non-GPU tensor raises.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import MotifConvDesc, check

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3, 4


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda or t.dtype not in (torch.float32, torch.int32):
        raise RuntimeError("motif_amd ops need CUDA(ROCm) fp32/int32 tensors, got %s %s" % (t.device, t.dtype))
    return ctypes.c_void_p(t.data_ptr())


def require_device(t, what="motif_amd ops need CUDA(ROCm) tensors"):
    """The product path has no host route: refuse host tensors up front with a clear message."""
    if not t.is_cuda:
        raise RuntimeError(what)


def _p8(t):
    if not t.is_cuda or t.dtype != torch.uint8:
        raise RuntimeError("expected a CUDA(ROCm) uint8 tensor, got %s %s" % (t.device, t.dtype))
    return ctypes.c_void_p(t.data_ptr())


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _packed_ready():
    """Packed weights are cached per parameter version and then used from whichever HIP stream runs the layer (RAFT's side
    stream, the per-image trunk streams, clips in flight): the one-off pack kernel is waited for on the host, so no later
    user on another stream can read a half-written blob."""
    torch.cuda.current_stream().synchronize()


def _planar(t):
    """True if t is [N,C,H,W] with dense C,H,W (batch stride free)."""
    n, c, h, w = t.shape
    return t.stride(3) == 1 and t.stride(2) == w and t.stride(1) == h * w


# ----------------------------------------------------------------------------------------- conv engine
# Arithmetic of the convolution contractions (MotifConvDesc.mma): 0 = fp32 MFMA, 6 = fp32-equivalent 3-way bf16
# split on the bf16 matrix cores (6 products, fp32 accumulate), 3 = 2-way split, 1 = plain bf16, 7 = fp32-equivalent 2-way
# fp16 split (3 products) in the kernels that have that form (conv_wino.hip: 3x3 stride 1) and 6 everywhere else.
MMA_FP32, MMA_BF16X3, MMA_BF16X2, MMA_BF16, MMA_F16X2 = 0, 6, 3, 1, 7
_MMA_NAMES = {"fp32": MMA_FP32, "bf16x3": MMA_BF16X3, "bf16x2": MMA_BF16X2, "bf16": MMA_BF16, "f16x2": MMA_F16X2}
_default_mma = _MMA_NAMES[os.environ.get("MOTIF_MMA", "f16x2")]
_conv_mma = int(os.environ.get("MOTIF_CONV_MMA", str(_default_mma)))


def set_conv_mma(mode):
    """Select the arithmetic for eligible conv layers (3x3, stride 1, > 32 couts); plans re-pack on the next call."""
    global _conv_mma
    if mode not in (MMA_FP32, MMA_BF16X3, MMA_BF16X2, MMA_BF16, MMA_F16X2):
        raise ValueError("conv mma mode must be 0, 6, 3, 1 or 7")
    _conv_mma = mode


def get_conv_mma():
    return _conv_mma


# ----------------------------------------------------------------------------------------- range status word
# include/motif_hip.h "Range status word": the kernels of the two-part fp16 arithmetic OR bit 0 into a caller-owned device word when an
# operand left fp16's range.  The word belongs to a MODEL INSTANCE (VideoSRBaseModel owns one and reads it in ensure_finite()); while
# that instance launches kernels it is the current word here.  Python issues the launches of a forward sequentially, so a plain
# module-level "current" is exact also with several instances and several streams in one process.
_status = None


class range_status:
    """with ops.range_status(word): every launch inside reports into `word` (int32[1] on the GPU; None = no report)."""

    def __init__(self, word):
        if word is not None and not (word.is_cuda and word.dtype == torch.int32 and word.numel() >= 1):
            raise RuntimeError("the range status word is a CUDA(ROCm) int32 tensor")
        self.word = word

    def __enter__(self):
        global _status
        self._saved, _status = _status, self.word
        return self.word

    def __exit__(self, *exc):
        global _status
        _status = self._saved
        return False


def _status_ptr():
    return ctypes.c_void_p(_status.data_ptr()) if _status is not None else None


class arithmetic:
    """with ops.arithmetic("bf16x3"): the launches inside use that arithmetic (None = whatever set_mma selected) -- a per-call
    override for one model instance; the process-wide selection is untouched."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global _conv_mma, _siren_mma
        self._saved = (_conv_mma, _siren_mma)
        if self.name is not None:
            mode = _MMA_NAMES[self.name]
            _conv_mma = mode
            _siren_mma = MMA_FP32 if mode == MMA_FP32 else MMA_F16X2 if mode == MMA_F16X2 else MMA_BF16X3
        return self

    def __exit__(self, *exc):
        global _conv_mma, _siren_mma
        _conv_mma, _siren_mma = self._saved
        return False


def set_option(name, value):
    """Tuning / test switch of the device library (include/motif_hip.h: motif_set_option); 0 = library default."""
    check(_lib.load().motif_set_option(name.encode(), int(value)), "motif_set_option(%s)" % name)


def get_option(name):
    v = ctypes.c_int(0)
    check(_lib.load().motif_get_option(name.encode(), ctypes.byref(v)), "motif_get_option(%s)" % name)
    return v.value


class ConvPlan:
    """Packed weights of one convolution layer, re-packed when the parameter (or the mma mode) changes."""

    def __init__(self, weight, bias, stride=1, pad=0, dil=1, groups=1, pad_mode=0, mma=None):
        self.weight, self.bias = weight, bias
        self.stride, self.pad, self.dil, self.groups, self.pad_mode = stride, pad, dil, groups, pad_mode
        self.mma = mma                     # None: follow set_conv_mma()
        self._packed = None
        self._key = None

    def desc(self, n, h, w, c0, c1=0):
        co, _, kh, kw = self.weight.shape
        d = MotifConvDesc()
        d.N, d.H, d.W, d.C0, d.C1 = n, h, w, c0, c1
        d.Cout, d.KH, d.KW = co, kh, kw
        d.stride, d.pad, d.dil, d.groups, d.pad_mode = self.stride, self.pad, self.dil, self.groups, self.pad_mode
        d.mma = _conv_mma if self.mma is None else self.mma
        return d

    def packed(self):
        w = self.weight
        key = (w.data_ptr(), w._version, w.device, _conv_mma if self.mma is None else self.mma)
        if self._key != key:
            lib = _lib.load()
            d = self.desc(1, 64, 64, w.shape[1] * self.groups)
            size = lib.motif_conv2d_packed_size(ctypes.byref(d))
            if size <= 0:
                raise RuntimeError("motif_conv2d_packed_size failed (%d) for weight %s" % (size, tuple(w.shape)))
            buf = torch.empty(size, dtype=torch.float32, device=w.device)
            wc = _c(w.detach())
            check(lib.motif_conv2d_pack(ctypes.byref(d), _p(wc), _p(buf), _stream()), "motif_conv2d_pack")
            _packed_ready()
            self._packed, self._key = buf, key
        return self._packed


def conv2d(plan, x, x2=None, act=ACT_NONE, res=None, res_mode=0, act2=ACT_NONE, act_split=0, out=None):
    """out = epilogue(conv(cat(x, x2))) through the fp32-MFMA implicit GEMM."""
    lib = _lib.load()
    if not _planar(x):
        x = x.contiguous()
    if x2 is not None and not _planar(x2):
        x2 = x2.contiguous()
    n, c0, h, w = x.shape
    c1 = x2.shape[1] if x2 is not None else 0
    d = plan.desc(n, h, w, c0, c1)
    d.act, d.act2, d.act_split, d.res_mode = act, act2, act_split, res_mode
    d.status = _status_ptr()
    kh, kw = plan.weight.shape[2:]
    ho = (h + 2 * plan.pad - (plan.dil * (kh - 1) + 1)) // plan.stride + 1
    wo = (w + 2 * plan.pad - (plan.dil * (kw - 1) + 1)) // plan.stride + 1
    co = plan.weight.shape[0]
    if out is None:
        out = torch.empty(n, co, ho, wo, dtype=torch.float32, device=x.device)
    elif not _planar(out) or tuple(out.shape) != (n, co, ho, wo):
        raise RuntimeError("conv2d: bad `out` view")
    d.in0_bs = x.stride(0)
    d.in1_bs = x2.stride(0) if x2 is not None else 0
    d.out_bs = out.stride(0)
    if res is not None:
        if not _planar(res):
            res = res.contiguous()
        d.res_bs = res.stride(0)
    bias = plan.bias.detach() if plan.bias is not None else None
    check(lib.motif_conv2d_fwd(ctypes.byref(d), _p(x), _p(x2), _p(plan.packed()), _p(bias), _p(res), _p(out), _stream()),
          "motif_conv2d_fwd")
    return out


CONV_CHAIN = os.environ.get("MOTIF_CONV_CHAIN", "1") != "0"     # residual chains as ONE persistent launch where the shape allows it
CONV_CHAIN_MIN_TILES = int(os.environ.get("MOTIF_CONV_CHAIN_MIN_TILES", "0"))      # 0: one tile per CU and layer (256 on an MI355X)
CONV_CHAIN_BYTES = int(os.environ.get("MOTIF_CONV_CHAIN_BYTES", str(280 << 20)))      # working set of one chain launch (three rotating buffers)


class conv_chain:
    """with ops.conv_chain(False): the residual trunks inside are launched layer by layer (same bits) -- what a model instance selects
    when one of its chain launches was abandoned (VideoSRBaseModel.ensure_finite, status bit 1).  None = leave the setting."""

    def __init__(self, enabled):
        self.enabled = enabled

    def __enter__(self):
        global CONV_CHAIN
        self._saved = CONV_CHAIN
        if self.enabled is not None:
            CONV_CHAIN = bool(self.enabled) and self._saved
        return self

    def __exit__(self, *exc):
        global CONV_CHAIN
        CONV_CHAIN = self._saved
        return False


def resblock_chain(blocks, x, out=None, act=ACT_RELU, last_act=ACT_NONE):
    """x' = x + conv2(act(conv1(x))) over `blocks` = [(plan1, plan2), ...] (`module_util.py:34-52` in an nn.Sequential: the reconstruction
    trunk and the feature extraction, `Ours.py:349-356`; with act = leaky ReLU and `last_act` applied to the LAST block's sum, the five
    LateralBlocks of `flow_process`, `Ours.py:425-431`).  One persistent launch (`motif_conv2d_chain_fwd`: the tiles of all layers in
    dependency order, conv_wino.hip CHAIN) where the shape and the arithmetic allow it -- the same bits as the launches layer by layer,
    which is what runs otherwise.  x [N,C,H,W] planar; `out` may be a batch-strided view."""
    lib = _lib.load()
    if not _planar(x):
        x = x.contiguous()
    n, c, h, w = x.shape
    if out is None:
        out = torch.empty(n, c, h, w, dtype=torch.float32, device=x.device)
    elif not _planar(out) or tuple(out.shape) != (n, c, h, w):
        raise RuntimeError("resblock_chain: bad `out` view")
    L = 2 * len(blocks)
    d = blocks[0][0].desc(n, h, w, c)
    ok = CONV_CHAIN and x.is_cuda and all(tuple(p.weight.shape) == (c, c, 3, 3) and p.stride == 1 and p.pad == 1 and p.dil == 1 and p.groups == 1 and
                                          p.pad_mode == 0 and (p.mma is None or p.mma == d.mma) for blk in blocks for p in blk)
    # below one tile per CU and layer the chain is bound by its dependency latency (a tile time per layer, like the launches) and gains nothing
    ok = ok and n * ((h + 7) // 8) * ((w + 31) // 32) >= (CONV_CHAIN_MIN_TILES or torch.cuda.get_device_properties(x.device).multi_processor_count)
    words = lib.motif_conv2d_chain_ws_words(ctypes.byref(d), L) if ok else 0
    if words <= 0 or ((x.data_ptr() | out.data_ptr()) & 15) or ((x.stride(0) | out.stride(0)) & 3):
        y = x
        for i, (p1, p2) in enumerate(blocks):
            lastb = i == len(blocks) - 1
            y = conv2d(p2, conv2d(p1, y, act=act), act=last_act if lastb else ACT_NONE, res=y, res_mode=1, out=out if lastb else None)
        return y
    # the chain's three rotating buffers (+ input / output) live in the 256 MB memory-side cache while they fit: 9 images of 64 x 180 x 320
    # as one chain took 1.11 ms per image against 0.96 for 3 or 6 (tools/chain_time.py) -- larger batches run as several chains
    per = 3 * c * h * w * 4
    nmax = max(1, CONV_CHAIN_BYTES // per)
    if n > nmax:
        for i0 in range(0, n, nmax):
            i1 = min(n, i0 + nmax)
            di = blocks[0][0].desc(i1 - i0, h, w, c)
            wi = lib.motif_conv2d_chain_ws_words(ctypes.byref(di), L)
            conv2d_chain(blocks, x[i0:i1], out[i0:i1], act, di, wi, last_act)
        return out
    return conv2d_chain(blocks, x, out, act, d, words, last_act)


def _chain_table(blocks, act, last_act, device):
    """MotifChainLayer[L] of a residual chain on the device.  The table holds the addresses of the layers' packed blobs and biases, so it
    lives exactly as long as they do: it is kept ON the first layer's plan (keyed by those addresses -- a re-pack or a new bias tensor makes
    a new table, the old one is released with the blob it pointed into) and never in a process-wide cache that could be cleared while a
    launch on another stream still reads it.  Built with a host wait (`_packed_ready`), so any stream may use it afterwards."""
    packed = [p.packed() for blk in blocks for p in blk]
    key = (tuple(t.data_ptr() for t in packed), tuple(p.bias.data_ptr() if p.bias is not None else 0 for blk in blocks for p in blk), act, last_act, str(device))
    owner = blocks[0][0]
    tabs = owner.__dict__.setdefault("_chain_tabs", {})
    tab = tabs.get(key)
    if tab is None:
        # MotifChainLayer[L]: packed, bias, src, dst, res, act | res_mode << 8.  Buffers: 0 = x, 1 = out, 2 = T, 3 / 4 = X (x_b lives in X[b % 2]):
        # conv1 of block b reads x_b and writes T; conv2 reads T (+ x_b) and writes x_b+1 -- the rotation the header proves hazard-free
        rows = []
        nb = len(blocks)
        for b, (p1, p2) in enumerate(blocks):
            xb = 0 if b == 0 else 3 + (b % 2)
            xn = 1 if b == nb - 1 else 3 + ((b + 1) % 2)
            for p, src, dst, res, arm in ((p1, xb, 2, -1, act), (p2, 2, xn, xb, (last_act if b == nb - 1 else ACT_NONE) | (1 << 8))):
                bias = p.bias.detach().data_ptr() if p.bias is not None else 0
                rows.append((p.packed().data_ptr(), bias, (src & 0xffffffff) | ((dst & 0xffffffff) << 32), (res & 0xffffffff) | ((arm & 0xffffffff) << 32)))
        import numpy as np
        tab = torch.from_numpy(np.array(rows, dtype=np.uint64).view(np.int64)).to(device)
        _packed_ready()
        if len(tabs) >= 4:                               # re-packed several times (new weights, another arithmetic): old tables go -- after a
            torch.cuda.synchronize(device)               # DEVICE-wide wait, since launches of any stream may still be reading them
            tabs.clear()
        tabs[key] = (tab, packed)                        # (the blobs are referenced so that the addresses stay theirs while the table exists)
    else:
        tab = tab[0]
    return tab


_chain_status = {}


def _chain_status_word(device):
    """A chain launch always reports: without a caller's status word (ops.range_status) it gets this per-device word, and
    `check_chain_status` raises if a launch was abandoned (tools, smoke(); a launch without ANY word traps on the device instead)."""
    w = _chain_status.get(str(device))
    if w is None:
        w = _chain_status[str(device)] = torch.zeros(1, dtype=torch.int32, device=device)
    return w


def check_chain_status(device=None):
    """Read (and clear) the fallback status words of chain launches made outside an ops.range_status context; raises RuntimeError if one
    of them was abandoned (bit 1) or carried a bad layer table (bit 2).  One host synchronisation per word."""
    for key, w in _chain_status.items():
        if device is not None and key != str(device):
            continue
        v = int(w.item())
        if v:
            w.zero_()
            raise RuntimeError("a chain launch on %s reported status %d (bit 1: abandoned after a second without progress, bit 2: bad layer table): its outputs are invalid" % (key, v))


def conv2d_chain(blocks, x, out, act, d, words, last_act=ACT_NONE):
    """The launch of `resblock_chain` (`motif_conv2d_chain_fwd`); eligibility was checked there."""
    lib = _lib.load()
    n, c, h, w = x.shape
    L = 2 * len(blocks)
    tab = _chain_table(blocks, act, last_act, x.device)
    work = torch.empty(3 * n * c * h * w, dtype=torch.float32, device=x.device)           # T, X[0], X[1]
    ws = torch.empty(words, dtype=torch.int32, device=x.device)
    d.in0_bs, d.out_bs = x.stride(0), out.stride(0)
    d.status = _status_ptr() if _status is not None else ctypes.c_void_p(_chain_status_word(x.device).data_ptr())
    check(lib.motif_conv2d_chain_fwd(ctypes.byref(d), L, ctypes.c_void_p(tab.data_ptr()), _p(x), _p(out), _p(work), work.numel(), _p(ws), _stream()), "motif_conv2d_chain_fwd")
    return out


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


def conv2d_multi(plans, xs, x2s=None, act=ACT_NONE, ress=None, res_mode=0, act2=ACT_NONE, act_split=0):
    """P <= 4 independent convolutions of identical shape in one launch -> stacked [P,N,Co,Ho,Wo]."""
    lib = _lib.load()
    P = len(plans)
    xs = [x if _planar(x) else x.contiguous() for x in xs]
    if x2s is not None:
        x2s = [x if _planar(x) else x.contiguous() for x in x2s]
    if ress is not None:
        ress = [x if _planar(x) else x.contiguous() for x in ress]
    n, c0, h, w = xs[0].shape
    c1 = x2s[0].shape[1] if x2s is not None else 0
    p0 = plans[0]
    d = p0.desc(n, h, w, c0, c1)
    d.act, d.act2, d.act_split, d.res_mode = act, act2, act_split, res_mode
    d.status = _status_ptr()
    kh, kw = p0.weight.shape[2:]
    ho = (h + 2 * p0.pad - (p0.dil * (kh - 1) + 1)) // p0.stride + 1
    wo = (w + 2 * p0.pad - (p0.dil * (kw - 1) + 1)) // p0.stride + 1
    co = p0.weight.shape[0]
    out = torch.empty(P, n, co, ho, wo, dtype=torch.float32, device=xs[0].device)
    outs = list(out)
    packs = [pl.packed() for pl in plans]
    biases = [pl.bias.detach() if pl.bias is not None else None for pl in plans]
    longs = lambda ts: (ctypes.c_long * P)(*[t.stride(0) for t in ts])
    check(lib.motif_conv2d_fwd_multi(
        ctypes.byref(d), P, _ptr_array(xs), _ptr_array(x2s) if x2s is not None else None, _ptr_array(packs),
        _ptr_array(biases) if biases[0] is not None else None, _ptr_array(ress) if ress is not None else None, _ptr_array(outs),
        longs(xs), longs(x2s) if x2s is not None else None, longs(ress) if ress is not None else None, longs(outs), _stream()),
        "motif_conv2d_fwd_multi")
    return out


def dcn_v2_multi(dplans, xs, oms, dg=8, act=ACT_NONE, kh=3, kw=3, stride=1, pad=1, dil=1):
    """P independent DCN_sep forwards (offset/mask tensors `oms` as produced by the fused conv_offset_mask)."""
    lib = _lib.load()
    P = len(dplans)
    xs = [x if _planar(x) else x.contiguous() for x in xs]
    oms = [_c(o) for o in oms]
    b, c, h, w = xs[0].shape
    co = dplans[0].weight.shape[0]
    ho = (h + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    wo = (w + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    t = kh * kw
    out = torch.empty(P, b, co, ho, wo, dtype=torch.float32, device=xs[0].device)
    outs = list(out)
    fused = (kh == 3 and kw == 3 and stride == 1 and pad == 1 and dil == 1 and (c // dg) % 4 == 0
             and act in (ACT_NONE, ACT_LRELU, ACT_RELU) and not os.environ.get("MOTIF_DCN_UNFUSED"))
    if fused:
        split = _conv_mma in (MMA_BF16X3, MMA_F16X2) and not os.environ.get("MOTIF_DCN_FP32")
        packs = [dp.packed_split() if split else dp.plan3x3().packed() for dp in dplans]
        biases = [dp.bias.detach() for dp in dplans]
        masks = (ctypes.c_void_p * P)(*[o.data_ptr() + 4 * 2 * dg * t * ho * wo for o in oms])
        bs = oms[0].stride(0)
        check(lib.motif_dcn_v2_fused_fwd_multi(P, _ptr_array(xs), (ctypes.c_long * P)(*[x.stride(0) for x in xs]), _ptr_array(oms), masks,
                                               _ptr_array(packs), _ptr_array(biases), _ptr_array(outs), b, c, h, w, co, dg, bs, bs, act,
                                               (_conv_mma if split else MMA_FP32), _status_ptr(), _stream()), "motif_dcn_v2_fused_fwd_multi")     # 6: three bf16 parts, 7: two fp16 parts in the window kernel
        return out
    cols = workspace(P * b * c * t * ho * wo, xs[0].device, "dcn_cols")
    plans = [dp.plan() for dp in dplans]
    packs = [pl.packed() for pl in plans]
    biases = [dp.bias.detach() for dp in dplans]
    masks = (ctypes.c_void_p * P)(*[o.data_ptr() + 4 * 2 * dg * t * ho * wo for o in oms])
    bs = oms[0].stride(0)
    check(lib.motif_dcn_v2_fwd_multi(P, _ptr_array(xs), (ctypes.c_long * P)(*[x.stride(0) for x in xs]), _ptr_array(oms), masks,
                                     _ptr_array(packs), _ptr_array(biases), _p(cols), _ptr_array(outs), b, c, h, w, co, kh, kw,
                                     stride, pad, dil, dg, bs, bs, act, _stream()), "motif_dcn_v2_fwd_multi")
    return out


class DcnPlan:
    """DCNv2 main weight viewed as a 1x1 conv over C*kh*kw column channels."""

    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias
        self._plan, self._key = None, None

    def plan3x3(self):
        """The weight as an ordinary 3x3 conv plan (K order (channel pair, tap, half)) for the fused DCN kernel."""
        if getattr(self, "_p3", None) is None or self._p3.weight is not self.weight:
            self._p3 = ConvPlan(self.weight, self.bias, 1, 1, 1, 1, 0, mma=MMA_FP32)   # fused DCN reads the fp32 format
        return self._p3

    def packed_split(self):
        """Weights split into bf16 A fragments for the fused kernel's mma = 6 form (motif_dcn_split_pack)."""
        w = self.weight
        key = (w.data_ptr(), w._version, w.device)
        if getattr(self, "_skey", None) != key:
            lib = _lib.load()
            co, ci = w.shape[0], w.shape[1]
            n = lib.motif_dcn_split_pack(None, None, co, ci, None)
            if n <= 0:
                raise RuntimeError("motif_dcn_split_pack size query failed (%d)" % n)
            buf = torch.empty(n, dtype=torch.float32, device=w.device)
            rc = lib.motif_dcn_split_pack(_p(_c(w.detach())), _p(buf), co, ci, _stream())
            if rc != n:
                raise RuntimeError("motif_dcn_split_pack failed (%d)" % rc)
            _packed_ready()
            self._spacked, self._skey = buf, key
        return self._spacked

    def plan(self):
        w = self.weight
        key = (w.data_ptr(), w._version, w.device)
        if self._key != key:
            co, ci, kh, kw = w.shape
            self._w1x1 = w.detach().reshape(co, ci * kh * kw, 1, 1)
            self._plan = ConvPlan(self._w1x1, self.bias)
            self._key = key
        return self._plan


_ws = {}


_ws_owner = None          # set by a model instance while it records a HIP graph (VideoSRBaseModel._test_graph)


def set_workspace_owner(owner):
    """Scratch recorded into a HIP graph is baked into it: every model instance captures on torch's ONE capture stream, so the
    stream alone would hand two instances the same buffers and their graphs, replayed concurrently, would race on them.
    While an instance records, its id is part of the key."""
    global _ws_owner
    _ws_owner = owner


def workspace(numel, device, tag="default"):
    """Scratch buffer per (tag, device, current stream[, recording model]): clips in flight on different streams never share one."""
    key = (tag, str(device), torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0, _ws_owner)
    buf = _ws.get(key)
    if buf is None or buf.numel() < numel:
        buf = torch.empty(numel, dtype=torch.float32, device=device)
        _ws[key] = buf
    return buf


def dcn_v2(dplan, x, offset_mask, dg=8, act=ACT_NONE, kh=3, kw=3, stride=1, pad=1, dil=1):
    """x [B,C,H,W]; offset_mask [B, 3*dg*kh*kw, Ho, Wo] = conv_offset_mask output with the mask third
    already sigmoid'ed (chunk/cat of dcn_v2.py:131-138 is a no-op on the channel order).  Same kernels as the
    multi-problem form (fused deformable im2col + MFMA for the 3x3/s1/p1 configuration)."""
    return dcn_v2_multi([dplan], [x], [offset_mask], dg, act, kh, kw, stride, pad, dil)[0]


def dcn_v2_raw(x, offset, mask, weight, bias, kh, kw, stride, pad, dil, dg, act=ACT_NONE):
    """Operator form of _ext.dcn_v2_forward (separate offset / mask tensors)."""
    lib = _lib.load()
    x, offset, mask = _c(x), _c(offset), _c(mask)
    b, c, h, w = x.shape
    co = weight.shape[0]
    ho = (h + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    wo = (w + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    cols = workspace(b * c * kh * kw * ho * wo, x.device, "dcn_cols")
    out = torch.empty(b, co, ho, wo, dtype=torch.float32, device=x.device)
    plan = ConvPlan(weight.detach().reshape(co, c * kh * kw, 1, 1), bias)
    check(lib.motif_dcn_v2_fwd(_p(x), _p(offset), _p(mask), _p(plan.packed()), _p(bias.detach() if bias is not None else None),
                               _p(cols), _p(out), b, c, h, w, co, kh, kw, stride, pad, dil, dg, 0, 0, act, _stream()),
          "motif_dcn_v2_fwd")
    return out


# ----------------------------------------------------------------------------------------- SIREN
def siren_pack(linears):
    """linears: list of (weight [out,in], bias [out]) device tensors -> packed blob tensor."""
    lib = _lib.load()
    n = len(linears)
    ws = [_c(w.detach()) for w, _ in linears]
    bs = [_c(b.detach()) for _, b in linears]
    dims = [ws[0].shape[1]] + [w.shape[0] for w in ws]
    wp = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
    bp = (ctypes.c_void_p * n)(*[b.data_ptr() for b in bs])
    dm = (ctypes.c_int * (n + 1))(*dims)
    total = lib.motif_siren_pack(wp, bp, dm, n, None, None)
    if total <= 0:
        raise RuntimeError("motif_siren_pack size query failed (%d)" % total)
    blob = torch.empty(total, dtype=torch.float32, device=ws[0].device)
    rc = lib.motif_siren_pack(wp, bp, dm, n, _p(blob), _stream())
    if rc != total:
        raise RuntimeError("motif_siren_pack failed (%d)" % rc)
    _packed_ready()
    return blob


SIREN_IMNET, SIREN_FLOW, SIREN_SYNTH, SIREN_SYNTH_PRE = 0, 1, 2, 3
_siren_mma = int(os.environ.get("MOTIF_SIREN_MMA", str(MMA_FP32 if _default_mma == MMA_FP32 else MMA_F16X2 if _default_mma == MMA_F16X2 else MMA_BF16X3)))


def set_mma(name):
    """Arithmetic of the dense contractions on the path: "bf16x3" (fp32-equivalent 3-way bf16 split on the bf16 matrix
    cores, convolutions and MLPs), "f16x2" (default: the same, with the fp32-equivalent 2-way fp16 split in the 3x3 stride-1
    convolutions conv_wino.hip serves), "fp32" (v_mfma_f32_32x32x2_f32 everywhere), "bf16x2" / "bf16" (convolutions only;
    reduced precision, the MLPs stay bf16x3)."""
    mode = _MMA_NAMES[name]
    set_conv_mma(mode)
    set_siren_mma(MMA_FP32 if mode == MMA_FP32 else MMA_F16X2 if mode == MMA_F16X2 else MMA_BF16X3)


def get_mma():
    return {v: k for k, v in _MMA_NAMES.items()}[_conv_mma]


def set_siren_mma(mode):
    """0: fp32-MFMA SIREN kernels; 6: 3-way bf16 split on the bf16 matrix cores (needs the LR partial, `pre`); 7: 2-way fp16
    split (three products instead of six) in the same kernels."""
    global _siren_mma
    if mode not in (MMA_FP32, MMA_BF16X3, MMA_F16X2):
        raise ValueError("siren mma mode must be 0, 6 or 7")
    _siren_mma = mode


def get_siren_mma():
    return _siren_mma


def siren_is_split():
    """The MLPs run on the 16-bit matrix cores (blobs from siren_pack_split)."""
    return _siren_mma in (MMA_BF16X3, MMA_F16X2)


def siren_pre():
    """`pre` argument of the SIREN forward calls for the selected arithmetic: 3 = two fp16 parts, 2 = three bf16 parts (blobs from
    siren_pack_split), 1 = fp32 MFMA on the LR partial (blob from siren_pack)."""
    return 3 if _siren_mma == MMA_F16X2 else 2 if _siren_mma == MMA_BF16X3 else 1


def siren_pack_split(kind, linears, pre=None):
    """Packed blob of one of the three MoTIF MLPs (kind = SIREN_IMNET / _FLOW / _SYNTH) for the split kernels; pass it
    with the same `pre` (2: three bf16 parts, 3: two fp16 parts; default: siren_pre() of the selected arithmetic)."""
    lib = _lib.load()
    pre = siren_pre() if pre is None else pre
    if pre not in (2, 3):
        raise RuntimeError("siren_pack_split: pre must be 2 or 3 (selected arithmetic: %d)" % _siren_mma)
    n = len(linears)
    ws = [_c(w.detach()) for w, _ in linears]
    bs = [_c(b.detach()) for _, b in linears]
    want = {SIREN_IMNET: [66, 64, 64, 256, 64], SIREN_FLOW: [67, 64, 64, 256, 3], SIREN_SYNTH: [198, 64, 64, 64, 256, 3],
            SIREN_SYNTH_PRE: [198, 64, 64, 64, 256, 3]}[kind]
    if [ws[0].shape[1]] + [w.shape[0] for w in ws] != want:
        raise RuntimeError("siren_pack_split: layer sizes do not match kind %d" % kind)
    wp = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
    bp = (ctypes.c_void_p * n)(*[b.data_ptr() for b in bs])
    kind = kind + (8 if pre == 3 else 0)
    total = lib.motif_siren_pack_split(kind, wp, bp, None, None)
    if total <= 0:
        raise RuntimeError("motif_siren_pack_split size query failed (%d)" % total)
    blob = torch.empty(total, dtype=torch.float32, device=ws[0].device)
    rc = lib.motif_siren_pack_split(kind, wp, bp, _p(blob), _stream())
    if rc != total:
        raise RuntimeError("motif_siren_pack_split failed (%d)" % rc)
    _packed_ready()
    return blob


def siren_imnet(blob, feat_lr, iy, ix, rel_y, rel_x, HH, WW, pre=False, add_lr=None):
    """add_lr [2B,64,H,W] (pre=2 only): gathered through the same tables and added to the output planes
    (motif_siren_imnet_add_fwd) -- the G term of the pre-contracted splat."""
    lib = _lib.load()
    feat_lr = _c(feat_lr)
    b2, c, h, w = feat_lr.shape
    out = torch.empty(b2, 64, HH, WW, dtype=torch.float32, device=feat_lr.device)
    if add_lr is not None:
        add_lr = _c(add_lr)
        if tuple(add_lr.shape) != (b2, 64, h, w):
            raise RuntimeError("siren_imnet: add_lr must be [%d,64,%d,%d]" % (b2, h, w))
        check(lib.motif_siren_imnet_add_fwd(_p(blob), _p(feat_lr), _p(add_lr), _p(iy), _p(ix), _p(rel_y), _p(rel_x), _p(out),
                                            b2, h, w, HH, WW, int(pre), _stream()), "motif_siren_imnet_add_fwd")
        return out
    check(lib.motif_siren_imnet_fwd(_p(blob), _p(feat_lr), _p(iy), _p(ix), _p(rel_y), _p(rel_x), _p(out),
                                    b2, h, w, HH, WW, int(pre), _stream()), "motif_siren_imnet_fwd")
    return out


def siren_flow(blob, flowfeat_lr, iy, ix, rel_y, rel_x, times, N, HH, WW, pre=False):
    lib = _lib.load()
    flowfeat_lr = _c(flowfeat_lr)
    b2, c, h, w = flowfeat_lr.shape
    pred = torch.empty(b2 * N, 3, HH, WW, dtype=torch.float32, device=flowfeat_lr.device)
    check(lib.motif_siren_flow_fwd(_p(blob), _p(flowfeat_lr), _p(iy), _p(ix), _p(rel_y), _p(rel_x), _p(_c(times)), _p(pred),
                                   b2, N, h, w, HH, WW, int(pre), _stream()), "motif_siren_flow_fwd")
    return pred


def _frames_out(out, N, B, HH, WW, device):
    if out is None:
        return torch.empty(N, B, 3, HH, WW, dtype=torch.float32, device=device)
    if tuple(out.shape) != (N, B, 3, HH, WW) or not out.is_contiguous():
        raise RuntimeError("frames `out` must be a contiguous [%d,%d,3,%d,%d] tensor" % (N, B, HH, WW))
    return out


def siren_synth(blob, acc, residual_lr, iy, ix, times, B, N, HH, WW, pre=False, out=None):
    lib = _lib.load()
    residual_lr = _c(residual_lr)
    _, _, h, w = residual_lr.shape
    frames = _frames_out(out, N, B, HH, WW, acc.device)
    check(lib.motif_siren_synth_fwd(_p(blob), _p(acc), _p(residual_lr), _p(iy), _p(ix), _p(_c(times)), _p(frames),
                                    B, N, h, w, HH, WW, int(pre), _status_ptr(), _stream()), "motif_siren_synth_fwd")
    return frames


def siren_synth_pre(blob, acc67, residual_l0, iy, ix, times, B, N, HH, WW, pre=None, out=None):
    """synth_net on the pre-contracted accumulator of splat_motif_pre (blob: siren_pack_split(SIREN_SYNTH_PRE, ...), same `pre`)."""
    lib = _lib.load()
    pre = siren_pre() if pre is None else pre
    residual_l0 = _c(residual_l0)
    _, _, h, w = residual_l0.shape
    frames = _frames_out(out, N, B, HH, WW, acc67.device)
    check(lib.motif_siren_synth_pre_fwd(_p(blob), _p(acc67), _p(residual_l0), _p(iy), _p(ix), _p(_c(times)), _p(frames),
                                        B, N, h, w, HH, WW, int(pre), _status_ptr(), _stream()), "motif_siren_synth_pre_fwd")
    return frames


def synth_input(acc, residual_lr, iy, ix, times, B, N, HH, WW):
    lib = _lib.load()
    residual_lr = _c(residual_lr)
    _, _, h, w = residual_lr.shape
    out = torch.empty(B * N, 198, HH, WW, dtype=torch.float32, device=acc.device)
    check(lib.motif_synth_input_fwd(_p(acc), _p(residual_lr), _p(iy), _p(ix), _p(_c(times)), _p(out), B, N, h, w, HH, WW, _stream()),
          "motif_synth_input_fwd")
    return out


# ----------------------------------------------------------------------------------------- splat
def splat(src, flow, z=None, want=("sum", "norm")):
    """Operator form.  Returns dict with the requested outputs among sum/norm/max/cnt."""
    lib = _lib.load()
    flow = _c(flow)
    n, _, h, w = flow.shape
    src = _c(src) if src is not None else None
    c = src.shape[1] if src is not None else 1
    dev = flow.device
    outs = {
        "sum": torch.zeros(n, c, h, w, device=dev) if "sum" in want else None,
        "norm": torch.zeros(n, 1, h, w, device=dev) if "norm" in want else None,
        "max": torch.ones(n, 1, h, w, device=dev) if "max" in want else None,
        "cnt": torch.zeros(n, 1, h, w, device=dev) if "cnt" in want else None,
    }
    check(lib.motif_splat_fwd(_p(src), _p(flow), _p(_c(z)) if z is not None else None, _p(outs["sum"]), _p(outs["norm"]),
                              _p(outs["max"]), _p(outs["cnt"]), n, c, h, w, _stream()), "motif_splat_fwd")
    return outs


def splat_motif(imnet_out, pred, feat_lr, iy, ix, alpha, flow_scale, B, N, HH, WW, acc=None, row0=0, accumulate=False):
    """Fused soft-splat of two source directions into acc [B*N,133,HH,WW]; accumulate=True adds a further pair of
    directions to an accumulator written by an earlier call (Ours_44 sums four)."""
    lib = _lib.load()
    feat_lr = _c(feat_lr)
    _, _, h, w = feat_lr.shape
    if acc is None:
        if accumulate:
            raise RuntimeError("splat_motif(accumulate=True) needs the accumulator of the first call")
        acc = torch.empty(B * N, 133, HH, WW, dtype=torch.float32, device=pred.device)
    # no zero fill: the owner-computes kernel writes every accumulator cell (max plane starts at 1)
    check(lib.motif_splat_motif_acc_fwd(_p(_c(imnet_out)), _p(_c(pred)), _p(feat_lr), _p(iy), _p(ix), _p(alpha.detach()), float(flow_scale), _p(acc),
                                        B, N, h, w, HH, WW, int(row0), int(bool(accumulate)), _stream()), "motif_splat_motif_acc_fwd")
    return acc


def splat_motif_pre(u_hr, pred, g_lr, ab, iy, ix, alpha, flow_scale, B, N, HH, WW, acc=None, row0=0, accumulate=False, lr_size=None):
    """Pre-contracted fused soft-splat (motif_splat_motif_pre_fwd): acc [B*N,67,HH,WW] = 64 first-layer pre-activation
    sums | norm | max | count.  g_lr=None: u_hr already holds U + G (siren_imnet(..., add_lr=g_lr) added the gathered LR term);
    lr_size=(H, W) is then required (the gather tables' LR size)."""
    lib = _lib.load()
    if g_lr is not None:
        g_lr = _c(g_lr)
        _, _, h, w = g_lr.shape
    else:
        if lr_size is None:
            raise RuntimeError("splat_motif_pre(g_lr=None) needs lr_size=(H, W)")
        h, w = lr_size
    if acc is None:
        if accumulate:
            raise RuntimeError("splat_motif_pre(accumulate=True) needs the accumulator of the first call")
        acc = torch.empty(B * N, 67, HH, WW, dtype=torch.float32, device=pred.device)
    check(lib.motif_splat_motif_pre_fwd(_p(_c(u_hr)), _p(_c(pred)), _p(g_lr) if g_lr is not None else None, _p(_c(ab)), _p(iy), _p(ix), _p(alpha.detach()),
                                        float(flow_scale), _p(acc), B, N, h, w, HH, WW, int(row0), int(bool(accumulate)), _stream()), "motif_splat_motif_pre_fwd")
    return acc


# ----------------------------------------------------------------------------------------- misc
def resize_bilinear(x, size, align_corners=False, mul=1.0, out=None, raft_norm=False):
    """raft_norm: the result is also normalised as RAFT's input (motif_resize_bilinear post = 1); out: a contiguous destination."""
    lib = _lib.load()
    x = _c(x)
    n, c, h, w = x.shape
    ho, wo = size
    if out is None:
        out = torch.empty(n, c, ho, wo, dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (n, c, ho, wo) or not out.is_contiguous():
        raise RuntimeError("resize_bilinear: `out` must be a contiguous [%d,%d,%d,%d] tensor" % (n, c, ho, wo))
    check(lib.motif_resize_bilinear(_p(x), _p(out), n * c, h, w, ho, wo, int(align_corners), float(mul), int(bool(raft_norm)), _stream()),
          "motif_resize_bilinear")
    return out


def backwarp(img, flow, sign=1.0):
    lib = _lib.load()
    img, flow = _c(img), _c(flow)
    n, c, h, w = img.shape
    out = torch.empty_like(img)
    check(lib.motif_backwarp(_p(img), _p(flow), _p(out), n, c, h, w, float(sign), _stream()), "motif_backwarp")
    return out


_linspace_cache = {}


def pwc_backward_warp(img, flow):
    lib = _lib.load()
    img, flow = _c(img), _c(flow)
    n, c, h, w = img.shape
    key = (h, w, str(img.device))
    if key not in _linspace_cache:
        _linspace_cache[key] = (torch.linspace(-1.0, 1.0, w).to(img.device), torch.linspace(-1.0, 1.0, h).to(img.device))
    gx, gy = _linspace_cache[key]
    out = torch.empty_like(img)
    check(lib.motif_pwc_backward_warp(_p(img), _p(flow), _p(gx), _p(gy), _p(out), n, c, h, w, _stream()), "motif_pwc_backward_warp")
    return out


def reliability(fr0, fr1, flow, g_filter, B, H, W):
    """fr0/fr1: [B,3,H,W] views (channel/row dense, any batch stride); flow [4B,2,H,W]."""
    lib = _lib.load()
    if not (_planar(fr0) and _planar(fr1) and fr0.stride(0) == fr1.stride(0)):
        fr0, fr1 = fr0.contiguous(), fr1.contiguous()
    flow = _c(flow)
    psies = torch.empty(4 * B, 3, H, W, dtype=torch.float32, device=flow.device)
    flow_feat = torch.empty(2 * B, 14, H, W, dtype=torch.float32, device=flow.device)
    check(lib.motif_reliability_fwd(_p(fr0), _p(fr1), fr0.stride(0), _p(flow), _p(_c(g_filter.detach())), _p(psies), _p(flow_feat),
                                    B, H, W, _stream()), "motif_reliability_fwd")
    return psies, flow_feat


def reliability_pairs(frames, flow, g_filter, table, durations, S):
    """Table-driven reliability maps + flow-encoder input of the 4-frame generators.  frames [B,n,3,H,W] (frame / batch
    strides free, planes dense); flow [F*B,2,H,W]; table: list of (src frame, dst frame, flow index, reverse flow index);
    durations: list of (d0, d1) per flow (already divided); S flows per source frame."""
    lib = _lib.load()
    B, n, _, H, W = frames.shape
    if not (frames.stride(4) == 1 and frames.stride(3) == W and frames.stride(2) == H * W):
        frames = frames.contiguous()
    flow = _c(flow)
    J = len(table)
    tab = (ctypes.c_int * (4 * J))(*[int(v) for row in table for v in row])
    dur = (ctypes.c_float * (2 * J))(*[float(v) for row in durations for v in row])
    psies = torch.empty(J * B, 3, H, W, dtype=torch.float32, device=flow.device)
    flow_feat = torch.empty((J // S) * B, S * 7, H, W, dtype=torch.float32, device=flow.device)
    check(lib.motif_reliability_pairs_fwd(_p(frames), frames.stride(1), frames.stride(0), _p(flow), _p(_c(g_filter.detach())), tab, dur, J, S,
                                          _p(psies), _p(flow_feat), B, H, W, _stream()), "motif_reliability_pairs_fwd")
    return psies, flow_feat


def instance_norm(x, mode=0, res=None):
    lib = _lib.load()
    x = _c(x)
    n, c, h, w = x.shape
    out = torch.empty_like(x)
    ws = workspace((n * c * 130) * 2, x.device, "instnorm")         # fp64 scratch (NC*(2+128) doubles), raw bytes to the library
    check(lib.motif_instance_norm_ws(_p(x), _p(_c(res)) if res is not None else None, _p(out), ctypes.c_void_p(ws.data_ptr()),
                                     n * c, h * w, mode, _stream()), "motif_instance_norm_ws")
    return out


def instance_norm_affine(x, gamma, beta):
    """torch.nn.InstanceNorm2d(C, affine=True) (PWCNet_light.py:18): per (image, channel) plane normalisation, then * gamma[c] + beta[c]."""
    lib = _lib.load()
    x = _c(x)
    n, c, h, w = x.shape
    out = torch.empty_like(x)
    ws = workspace((n * c * 130) * 2, x.device, "instnorm")
    check(lib.motif_instance_norm_affine_ws(_p(x), _p(_c(gamma.detach())), _p(_c(beta.detach())), _p(out), ctypes.c_void_p(ws.data_ptr()),
                                            n, c, h * w, _stream()), "motif_instance_norm_affine_ws")
    return out


def instance_norm_synced(x, rows, reduce_sum, mode=0, res=None):
    """InstanceNorm with statistics over several processes: moments over this rank's own rows `rows` = (lo, hi) of every plane,
    `reduce_sum(t)` all-reduces the [N*C,3] float64 tensor in place (sum, sum of squares, count), then the whole plane is
    normalised with the global mean / variance."""
    lib = _lib.load()
    x = _c(x)
    n, c, h, w = x.shape
    sums = torch.empty(n * c, 3, dtype=torch.float64, device=x.device)
    check(lib.motif_instance_norm_moments(_p(x), ctypes.c_void_p(sums.data_ptr()), n * c, h, w, int(rows[0]), int(rows[1]), _stream()),
          "motif_instance_norm_moments")
    reduce_sum(sums)
    out = torch.empty_like(x)
    check(lib.motif_instance_norm_apply(_p(x), _p(_c(res)) if res is not None else None, ctypes.c_void_p(sums.data_ptr()), _p(out),
                                        n * c, h * w, mode, _stream()), "motif_instance_norm_apply")
    return out


def avg_pool2(x):
    lib = _lib.load()
    x = _c(x)
    n, c, h, w = x.shape
    out = torch.empty(n, c, h // 2, w // 2, dtype=torch.float32, device=x.device)
    check(lib.motif_avg_pool2(_p(x), _p(out), n * c, h, w, _stream()), "motif_avg_pool2")
    return out


def nchw_to_nhwc(x):
    lib = _lib.load()
    x = _c(x)
    n, c, h, w = x.shape
    out = torch.empty(n, h, w, c, dtype=torch.float32, device=x.device)
    check(lib.motif_nchw_to_nhwc(_p(x), _p(out), n, c, h * w, _stream()), "motif_nchw_to_nhwc")
    return out


def raft_corr_lookup(fmap1_nhwc, fmap2_nhwc, coords, coord_scale, out, ch_off, div, r=3):
    lib = _lib.load()
    b, h1, w1, c = fmap1_nhwc.shape
    _, h2, w2, _ = fmap2_nhwc.shape
    check(lib.motif_raft_corr_lookup(_p(fmap1_nhwc), _p(fmap2_nhwc), _p(_c(coords)), float(coord_scale), _p(out),
                                     b, h1, w1, h2, w2, c, r, out.shape[1], ch_off, float(div), _stream()), "motif_raft_corr_lookup")
    return out


def raft_corr_lookup_pyramid(fmap1_nhwc, fmap2_levels, coords, out, div, r=3, index1=None, index2=None):
    """index1 / index2 (host int lists, one entry per pair = row of `coords`): pair i correlates fmap1[index1[i]] with
    fmap2[.][index2[i]]; None = the maps are already ordered per pair."""
    lib = _lib.load()
    _, h1, w1, c = fmap1_nhwc.shape
    b = coords.shape[0]
    n = len(fmap2_levels)
    hs = (ctypes.c_int * n)(*[f.shape[1] for f in fmap2_levels])
    ws = (ctypes.c_int * n)(*[f.shape[2] for f in fmap2_levels])
    for idx, nmaps in ((index1, fmap1_nhwc.shape[0]), (index2, fmap2_levels[0].shape[0])):
        if idx is not None and (len(idx) != b or min(idx) < 0 or max(idx) >= nmaps):
            raise RuntimeError("raft_corr_lookup_pyramid: bad batch index map %r for %d pairs over %d maps" % (list(idx), b, nmaps))
    if index1 is None and index2 is None and (fmap1_nhwc.shape[0] != b or fmap2_levels[0].shape[0] != b):
        raise RuntimeError("raft_corr_lookup_pyramid: %d pairs but %d / %d feature maps" % (b, fmap1_nhwc.shape[0], fmap2_levels[0].shape[0]))
    i1 = (ctypes.c_int * b)(*[int(v) for v in index1]) if index1 is not None else None
    i2 = (ctypes.c_int * b)(*[int(v) for v in index2]) if index2 is not None else None
    check(lib.motif_raft_corr_lookup_pyramid(_p(fmap1_nhwc), _ptr_array(fmap2_levels), hs, ws, n, _p(_c(coords)), _p(out),
                                             b, h1, w1, c, r, out.shape[1], float(div), i1, i2, _stream()), "motif_raft_corr_lookup_pyramid")
    return out


def corr81(first, second, act=ACT_NONE, out=None):
    lib = _lib.load()
    first, second = _c(first), _c(second)
    b, c, h, w = first.shape
    if out is None:
        out = torch.empty(b, 81, h, w, dtype=torch.float32, device=first.device)
    elif tuple(out.shape) != (b, 81, h, w) or not out.is_contiguous():
        raise RuntimeError("corr81: `out` must be a contiguous [%d,81,%d,%d] tensor" % (b, h, w))
    check(lib.motif_corr81_fwd(_p(first), _p(second), _p(out), b, c, h, w, act, _stream()), "motif_corr81_fwd")
    return out


def gru_update(z, q, h):
    lib = _lib.load()
    out = torch.empty_like(h)
    check(lib.motif_gru_update(_p(_c(z)), _p(_c(q)), _p(_c(h)), _p(out), h.numel(), _stream()), "motif_gru_update")
    return out


def lstm_gates(cc, c_cur):
    lib = _lib.load()
    cc, c_cur = _c(cc), _c(c_cur)
    b, c4, h, w = cc.shape
    h_next, c_next = torch.empty_like(c_cur), torch.empty_like(c_cur)
    check(lib.motif_lstm_gates(_p(cc), _p(c_cur), _p(h_next), _p(c_next), b, c4 // 4, h * w, _stream()), "motif_lstm_gates")
    return h_next, c_next


def axpby(x, y=None, a=1.0, b=1.0):
    lib = _lib.load()
    x = _c(x)
    out = torch.empty_like(x)
    check(lib.motif_axpby(_p(x), _p(_c(y)) if y is not None else None, float(a), float(b), _p(out), x.numel(), _stream()), "motif_axpby")
    return out


def axpby_into(x, y, a, b, out):
    """out[i] = a*x[i] + b*y[i] for the items i of the leading dimension, `out` a batch-strided view with dense items (a channel
    slice of a wider tensor): motif_axpby_bs."""
    lib = _lib.load()
    x = _c(x)
    n = x[0].numel()
    if tuple(out.shape) != tuple(x.shape) or not out[0].is_contiguous():
        raise RuntimeError("axpby_into: `out` must have x's shape and dense items")
    check(lib.motif_axpby_bs(_p(x), _p(_c(y)) if y is not None else None, float(a), float(b), _p(out), x.shape[0], n, out.stride(0), _stream()),
          "motif_axpby_bs")
    return out


def flow_roundtrip(pred, a, b):
    """pred [N,3,HH,WW] -> [N,2,HH,WW] = ((pred[:, :2] * a) * b / a) / b, the reference's scale-up / scale-back of the flow in one pass."""
    lib = _lib.load()
    pred = _c(pred)
    n, _, hh, ww = pred.shape
    out = torch.empty(n, 2, hh, ww, dtype=torch.float32, device=pred.device)
    check(lib.motif_flow_roundtrip(_p(pred), _p(out), n, hh * ww, float(a), float(b), _stream()), "motif_flow_roundtrip")
    return out


def deconv4x4s2(x, weight, bias, out=None):
    lib = _lib.load()
    x = _c(x)
    n, ci, h, w = x.shape
    co = weight.shape[1]
    if out is None:
        out = torch.empty(n, co, 2 * h, 2 * w, dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (n, co, 2 * h, 2 * w) or not out.is_contiguous():
        raise RuntimeError("deconv4x4s2: `out` must be a contiguous [%d,%d,%d,%d] tensor" % (n, co, 2 * h, 2 * w))
    check(lib.motif_deconv4x4s2(_p(x), _p(_c(weight.detach())), _p(bias.detach()) if bias is not None else None, _p(out),
                                n, ci, co, h, w, _stream()), "motif_deconv4x4s2")
    return out


# ----------------------------------------------------------------------------------------- frame formats
def frames_u8_to_f32(frames_u8, swap_rb=True):
    """uint8 [N,H,W,3] (cv2 order) -> fp32 [N,3,H,W] in [0,1] (data/Adobe_test_3.py:171-195)."""
    lib = _lib.load()
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[-1] != 3:
        raise RuntimeError("frames_u8_to_f32 expects uint8 [N,H,W,3]")
    x = frames_u8.contiguous()
    n, h, w, _ = x.shape
    out = torch.empty(n, 3, h, w, dtype=torch.float32, device=x.device)
    check(lib.motif_frames_u8_to_f32(_p8(x), _p(out), n, h, w, int(swap_rb), _stream()), "motif_frames_u8_to_f32")
    return out


def frames_f32_to_u8(frames, round_half_even=True, swap_rb=True):
    """fp32 [N,3,H,W] -> uint8 [N,H,W,3]: tensor2img (utils/util.py:105-129) by default, demo.py:94-99 with
    round_half_even=False, swap_rb=False."""
    lib = _lib.load()
    x = _c(frames)
    if x.dim() != 4 or x.shape[1] != 3:
        raise RuntimeError("frames_f32_to_u8 expects fp32 [N,3,H,W]")
    n, _, h, w = x.shape
    out = torch.empty(n, h, w, 3, dtype=torch.uint8, device=x.device)
    check(lib.motif_frames_f32_to_u8(_p(x), _p8(out), n, h, w, int(round_half_even), int(swap_rb), _stream()), "motif_frames_f32_to_u8")
    return out
