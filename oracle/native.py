"""ctypes binding of oracle/native_ref.c (test infrastructure; see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmotif_oracle.so")
_SRC = os.path.join(_HERE, "native_ref.c")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-ffp-contract=off",
                               _SRC, "-o", _SO, "-lm"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(t):
    assert t.dtype == torch.float32 and t.is_contiguous() and t.device.type == "cpu"
    return ctypes.c_void_p(t.data_ptr())


def splat(inp, flow, mode):
    """mode: 'sum' | 'max' | 'count'.  Mirrors _FunctionSoftsplat.forward of the three modules."""
    inp = inp.detach().float().contiguous()
    flow = flow.detach().float().contiguous()
    n, c, h, w = inp.shape
    assert flow.shape == (n, 2, h, w)
    m = {"sum": 0, "max": 1, "count": 2}[mode]
    out = torch.ones_like(inp) if m == 1 else torch.zeros_like(inp)
    lib().oracle_splat(_p(inp), _p(flow), _p(out), n, c, h, w, m)
    return out


def corr81(first, second):
    first = first.detach().float().contiguous()
    second = second.detach().float().contiguous()
    b, c, h, w = first.shape
    out = torch.zeros(b, 81, h, w)
    lib().oracle_corr81(_p(first), _p(second), _p(out), b, c, h, w)
    return out


def dcn_v2_forward(inp, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, dg):
    inp, weight, bias = (t.detach().float().contiguous() for t in (inp, weight, bias))
    offset, mask = offset.detach().float().contiguous(), mask.detach().float().contiguous()
    b, c, h, w = inp.shape
    co = weight.shape[0]
    ho = (h + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    wo = (w + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    out = torch.empty(b, co, ho, wo)
    lib().oracle_dcn_v2_forward(_p(inp), _p(weight), _p(bias), _p(offset), _p(mask), _p(out),
                                b, c, h, w, co, kh, kw, sh, sw, ph, pw, dh, dw, dg)
    return out


def alt_corr(fmap1, fmap2, coords, r):
    """alt_cuda_corr.forward(fmap1[B,H,W,C], fmap2[B,H2,W2,C], coords[B,1,H,W,2], r) -> (corr,)"""
    fmap1, fmap2, coords = (t.detach().float().contiguous() for t in (fmap1, fmap2, coords))
    b, h1, w1, c = fmap1.shape
    _, h2, w2, _ = fmap2.shape
    rd = 2 * r + 1
    out = torch.empty(b, 1, rd * rd, h1, w1)
    lib().oracle_alt_corr(_p(fmap1), _p(fmap2), _p(coords), _p(out), b, h1, w1, h2, w2, c, r)
    return (out,)
