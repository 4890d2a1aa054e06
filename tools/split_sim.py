#!/usr/bin/env python3
"""Representation error of the split arithmetics on a K = 576 contraction (numpy, products and sums in fp64 so that only the operand
splits and the dropped terms show): fp32 matmul, three bf16 parts / six products, two fp16 parts / three products with operand scales sx / sw."""
import numpy as np
rng = np.random.default_rng(0)
def bf16_trunc(x):
    return (x.astype(np.float32).view(np.uint32) & 0xffff0000).view(np.float32)
def split_bf16x3(x):
    x = x.astype(np.float32); p0 = bf16_trunc(x); r = x - p0; p1 = bf16_trunc(r); r2 = r - p1; p2 = bf16_trunc(r2)
    return [p0.astype(np.float64), p1.astype(np.float64), p2.astype(np.float64)]
def split_f16x2(x, s=1.0):
    x = (x.astype(np.float32) * np.float32(s))
    h = x.astype(np.float16); r = x - h.astype(np.float32); l = r.astype(np.float16)
    return [h.astype(np.float64) / s, l.astype(np.float64) / s]
def run(xs, ws, K=576, M=256, N=256):
    x = (rng.random((M, K)) * 2 - 1) * xs
    w = (rng.random((K, N)) * 2 - 1) * ws
    x = x.astype(np.float32); w = w.astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64)
    scale = np.abs(ref).mean()
    out = {}
    out["fp32"] = (x @ w).astype(np.float64)
    a = split_bf16x3(x); b = split_bf16x3(w)
    out["bf16x3"] = sum(a[i] @ b[j] for i, j in [(0,0),(0,1),(1,0),(1,1),(0,2),(2,0)])
    for sx, sw in [(1,1),(1,8),(0.125,8),(1,64),(1, 1024)]:
        a = split_f16x2(x, sx); b = split_f16x2(w, sw)
        out["f16x2 sx=%g sw=%g" % (sx, sw)] = a[0] @ b[0] + a[0] @ b[1] + a[1] @ b[0]
    print("x scale %g  w scale %g  mean|ref| %.3g" % (xs, ws, scale))
    for k, v in out.items():
        e = np.abs(v - ref)
        print("   %-22s max %.2e  rms %.2e  (rel to mean|ref|: max %.2e rms %.2e)" % (k, e.max(), np.sqrt((e**2).mean()), e.max()/scale, np.sqrt((e**2).mean())/scale))
run(1.0, 1/24)
run(1.0, 0.01)
run(0.05, 0.02)
run(30.0, 0.02)
run(1e-3, 0.02)
