#!/usr/bin/env python3
"""PWC-Net timing: the LDS-tiled 81-way cost volume against the round-1 kernel (one thread
per (displacement, pixel), tools/corr81_naive.hip) at the pyramid levels of a 768x1280 input, and one PWCNet forward."""
import ctypes
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from motif_amd import ops

HERE = os.path.dirname(os.path.abspath(__file__))


def naive_lib():
    so = os.path.join(HERE, "libcorr81_naive.so")
    if not os.path.exists(so):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(HERE, "corr81_naive.hip"), "-o", so])
    lib = ctypes.CDLL(so)
    lib.corr81_naive.restype = ctypes.c_int
    lib.corr81_naive.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    return lib


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / reps


def main():
    lib = naive_lib()
    B = 1
    print("level (C,H,W)        naive us   tiled us   speed-up   tiled GFLOP/s   max|diff|")
    for (C, H, W) in ((196, 12, 20), (128, 24, 40), (96, 48, 80), (64, 96, 160), (32, 192, 320)):
        a = torch.randn(B, C, H, W, device="cuda")
        b = torch.randn(B, C, H, W, device="cuda")
        out0 = torch.empty(B, 81, H, W, device="cuda")
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        t0 = timeit(lambda: lib.corr81_naive(a.data_ptr(), b.data_ptr(), out0.data_ptr(), B, C, H, W, st))
        ops.set_option("corr81", 1)
        t1 = timeit(lambda: ops.corr81(a, b))
        d = float((ops.corr81(a, b) - out0).abs().max())
        ops.set_option("corr81", 0)
        t2 = timeit(lambda: ops.corr81(a, b))                    # the library's own choice for this size
        print("(%3d,%3d,%3d)        %8.1f   %8.1f   %7.2fx   %10.1f      %.1e   auto: %.1f us" % (
            C, H, W, t0, t1, t0 / t1, 2.0 * B * 81 * C * H * W / t1 / 1e3, d, t2))
    from motif_amd.OpticalFlow.PWCNet import PWCNet
    from motif_amd.utils.synth_weights import fill_state_dict
    net = fill_state_dict(PWCNet()).cuda().eval()
    f0, f1 = torch.rand(1, 3, 720, 1280, device="cuda"), torch.rand(1, 3, 720, 1280, device="cuda")
    with torch.no_grad():
        t = timeit(lambda: net(f0, f1), reps=5)
    print("PWCNet forward, one 720x1280 pair: %.2f ms" % (t / 1000.0))


if __name__ == "__main__":
    main()
