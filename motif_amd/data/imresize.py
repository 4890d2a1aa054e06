"""MATLAB-style bicubic resize with antialiasing -- how the reference makes its LR frames.

Behaviour follows `data/util.py:261-400` of the reference (`imresize_np`, after MATLAB `imresize`): separable cubic kernel
(a = -0.5), widened by 1/scale when shrinking with antialiasing, output pixel k centred at (k + 0.5) / scale - 0.5, rows of
weights normalised to 1, symmetric extension at the borders, H first, then W, float32 arithmetic.  Built here as two small
matrices (one per axis) applied with a matmul instead of the reference's per-row loop; `tests/test_host.py` compares it with
fixtures produced by the reference function itself (`tests/golden/make_golden.py:imresize_case`)."""
import math

import numpy as np
import torch


def _cubic(x):
    ax = x.abs()
    ax2, ax3 = ax * ax, ax * ax * ax
    return (1.5 * ax3 - 2.5 * ax2 + 1) * (ax <= 1).to(x.dtype) + (-0.5 * ax3 + 2.5 * ax2 - 4 * ax + 2) * ((ax > 1) & (ax <= 2)).to(x.dtype)


def resize_matrix(in_len, scale, antialiasing=True):
    """[out_len, in_len] float32 matrix M with out = M @ in along one axis (symmetric border extension folded in)."""
    out_len = math.ceil(in_len * scale)
    shrink = scale < 1 and antialiasing
    kw = 4.0 / scale if shrink else 4.0
    x = torch.linspace(1, out_len, out_len)                               # 1-based output coordinates, float32 like the reference
    u = x / scale + 0.5 * (1 - 1 / scale)
    left = torch.floor(u - kw / 2)
    P = math.ceil(kw) + 2
    idx = left.view(-1, 1) + torch.linspace(0, P - 1, P).view(1, -1)      # 1-based input pixels of every output pixel
    d = u.view(-1, 1) - idx
    w = scale * _cubic(d * scale) if shrink else _cubic(d)
    w = w / w.sum(1, keepdim=True)
    # symmetric extension: pixel 0, -1, -2 ... -> 1, 2, 3 ...; in_len + 1, ... -> in_len, in_len - 1, ...
    i0 = idx.long() - 1
    period = 2 * in_len
    r = torch.remainder(i0, period)
    src = torch.where(r < in_len, r, period - 1 - r)
    M = torch.zeros(out_len, in_len, dtype=torch.float32)
    M.scatter_add_(1, src, w.float())
    return M


def imresize(img, scale, antialiasing=True):
    """img: numpy HWC (any channel count) or torch [..., H, W]; same container back, float32."""
    if isinstance(img, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(img)).float()
        mh, mw = resize_matrix(t.shape[0], scale, antialiasing), resize_matrix(t.shape[1], scale, antialiasing)
        out = torch.einsum("oh,hwc->owc", mh, t)
        out = torch.einsum("pw,owc->opc", mw, out)
        return out.numpy()
    t = img.float()
    mh, mw = resize_matrix(t.shape[-2], scale, antialiasing).to(t.device), resize_matrix(t.shape[-1], scale, antialiasing).to(t.device)
    return torch.einsum("pw,...ow->...op", mw, torch.einsum("oh,...hw->...ow", mh, t))


imresize_np = imresize      # the reference's name (data/util.py:323)
