#!/usr/bin/env python3
"""Representation error of the split arithmetics on a K = 576 contraction (numpy, products and sums in fp64 so that only the operand
splits and the dropped terms show): fp32 matmul, three bf16 parts / six products, and the two-part fp16 form in its round-4 shape
(plain low activation part) and its shipped round-5 shape (low activation part times 2^11, met by 2^-11 x the high weight part:
conv_wino.hip WOrder<2>, conv_pw.hip, dcn.hip, the first layers of siren_split.hip).  Weights of both fp16 forms are packed times 2^8."""
import numpy as np
rng = np.random.default_rng(0)
def bf16_trunc(x):
    return (x.astype(np.float32).view(np.uint32) & 0xffff0000).view(np.float32)
def split_bf16x3(x):
    x = x.astype(np.float32); p0 = bf16_trunc(x); r = x - p0; p1 = bf16_trunc(r); r2 = r - p1; p2 = bf16_trunc(r2)
    return [p0.astype(np.float64), p1.astype(np.float64), p2.astype(np.float64)]
def split_f16x2(x, s=1.0):
    """round 4: hi = rne(s x), lo = rne(s x - hi)"""
    x = (x.astype(np.float32) * np.float32(s))
    h = x.astype(np.float16); r = x - h.astype(np.float32); l = r.astype(np.float16)
    return [h.astype(np.float64) / s, l.astype(np.float64) / s]
def f16x2_shipped(x, w):
    """round 5, as the kernels compute it: x -> hi = rne(x), lo_s = rne((x - hi) 2^11);  W = 2^8 w -> Whi = rne(W), Wlo = rne(W - Whi),
    Whs = rne(2^-11 Whi) (v_pk_mul_f16);  out = 2^-8 (hi Whi + hi Wlo + lo_s Whs)"""
    x = x.astype(np.float32)
    hi = x.astype(np.float16)
    los = ((x - hi.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    W = w.astype(np.float32) * np.float32(256)
    Whi = W.astype(np.float16)
    Wlo = (W - Whi.astype(np.float32)).astype(np.float16)
    Whs = (Whi.astype(np.float32) * np.float32(2.0 ** -11)).astype(np.float16)
    f = lambda a: a.astype(np.float64)
    return (f(hi) @ f(Whi) + f(hi) @ f(Wlo) + f(los) @ f(Whs)) / 256.0
def run(xs, ws, K=576, M=256, N=256):
    x = ((rng.random((M, K)) * 2 - 1) * xs).astype(np.float32)
    w = ((rng.random((K, N)) * 2 - 1) * ws).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64)
    scale = np.abs(ref).mean()
    out = {}
    out["fp32"] = (x @ w).astype(np.float64)
    a = split_bf16x3(x); b = split_bf16x3(w)
    out["bf16x3"] = sum(a[i] @ b[j] for i, j in [(0,0),(0,1),(1,0),(1,1),(0,2),(2,0)])
    a = split_f16x2(x); b = split_f16x2(w, 256.0)
    out["f16x2 round 4 (plain lo)"] = a[0] @ b[0] + a[0] @ b[1] + a[1] @ b[0]
    out["f16x2 shipped (lo x 2^11)"] = f16x2_shipped(x, w)
    print("x scale %g  w scale %g  mean|ref| %.3g" % (xs, ws, scale))
    e32 = np.sqrt(((out["fp32"] - ref) ** 2).mean())
    for k, v in out.items():
        e = np.abs(v - ref)
        rms = np.sqrt((e**2).mean())
        print("   %-28s max %.2e  rms %.2e  (rel to mean|ref|: max %.2e rms %.2e;  rms / fp32's %.2f)" % (k, e.max(), rms, e.max()/scale, rms/scale, rms / e32))
if __name__ == "__main__":
    for xs in (1.0, 30.0, 1e4, 0.05, 1e-2, 1e-3, 1e-4, 1e-5):
        for ws in (1/24, 1e-2):
            run(xs, ws)
