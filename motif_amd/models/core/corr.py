"""RAFT correlation lookup on the HIP kernel that replaces `alt_cuda_corr`.

Mirrors `AlternateCorrBlock` (`/root/reference/models/core/corr.py:59-87`): 5-level avg-pool pyramid of
fmap2 (4 used), per level a windowed bilinear lookup at coords/2^i, stacked to [B,196,H,W], / sqrt(C).
`alt_cuda_corr_forward` keeps the third-party extension's call signature (corr.py:78-83).
"""
import math

import torch

from ... import ops


def alt_cuda_corr_forward(fmap1, fmap2, coords, r):
    """alt_cuda_corr.forward(fmap1[B,H,W,C], fmap2[B,H2,W2,C], coords[B,1,H,W,2], r) -> (corr[B,1,(2r+1)^2,H,W],)"""
    b, h, w, _ = fmap1.shape
    out = torch.empty(b, (2 * r + 1) ** 2, h, w, dtype=torch.float32, device=fmap1.device)
    c = coords.reshape(b, h, w, 2).permute(0, 3, 1, 2).contiguous()
    ops.raft_corr_lookup(fmap1.contiguous(), fmap2.contiguous(), c, 1.0, out, 0, 1.0, r)
    return (out.unsqueeze(1),)


class AlternateCorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4, index1=None, index2=None):
        """index1 / index2 (host int lists): the maps of pair i are fmap1[index1[i]] and fmap2[index2[i]] -- the channels-last
        copies and the avg-pool pyramid are built once per distinct map; the look-up kernel takes the pairing as two index maps
        (no gathered copies)."""
        self.num_levels = num_levels
        self.radius = radius
        self.dim = fmap1.shape[1]
        self.index1 = [int(v) for v in index1] if index1 is not None else None
        self.index2 = [int(v) for v in index2] if index2 is not None else None
        self.f1 = ops.nchw_to_nhwc(fmap1)
        self.f2 = []
        for i in range(self.num_levels):
            self.f2.append(ops.nchw_to_nhwc(fmap2))
            if i + 1 < self.num_levels:
                fmap2 = ops.avg_pool2(fmap2)
        self._div = float(torch.sqrt(torch.tensor(self.dim).float()))      # corr.py:87, once per block (a host computation)

    def __call__(self, coords):
        b, _, h, w = coords.shape
        n = (2 * self.radius + 1) ** 2
        out = torch.empty(b, self.num_levels * n, h, w, dtype=torch.float32, device=coords.device)
        return ops.raft_corr_lookup_pyramid(self.f1, self.f2, coords, out, self._div, self.radius, self.index1, self.index2)
