"""RAFT feature/context encoders on the HIP conv engine.

Mirrors `/root/reference/models/core/extractor.py:60-116` (BottleneckBlock) and `:195-267`
(SmallEncoder) -- same constructor arguments and state-dict keys; InstanceNorm has no parameters.
"""
import torch
import torch.nn as nn

import contextlib
import threading

from ... import ops
from ..modules.layers import Conv2d

_SYNC = threading.local()


@contextlib.contextmanager
def norm_sync(rows, total_rows, reduce_sum):
    """While active (this thread), the encoders' InstanceNorms take their statistics over ALL ranks of a row-tiled clip
    (SURVEY.md 8(e) row 3): this rank owns the rows [rows[0], rows[1]) of the `total_rows` rows of the encoder's input (its
    band without the context rows around it; the bounds must stay integers at 1/2, 1/4 and 1/8 resolution), `reduce_sum(t)`
    sums a float64 tensor over the ranks in place.  Every rank must run the same layers in the same order."""
    _SYNC.ctx = (int(rows[0]), int(rows[1]), int(total_rows), reduce_sum)
    try:
        yield
    finally:
        _SYNC.ctx = None


def _inorm(x, mode, res=None):
    ctx = getattr(_SYNC, "ctx", None)
    if ctx is None:
        return ops.instance_norm(x, mode, res=res)
    lo, hi, total, reduce_sum = ctx
    h = x.shape[2]
    if (lo * h) % total or (hi * h) % total:
        raise RuntimeError("norm_sync: rows (%d, %d) of %d do not map to whole rows of a %d-row map" % (lo, hi, total, h))
    return ops.instance_norm_synced(x, (lo * h // total, hi * h // total), reduce_sum, mode, res=res)


class BottleneckBlock(nn.Module):
    def __init__(self, in_planes, planes, norm_fn="group", stride=1):
        super().__init__()
        if norm_fn not in ("instance", "none"):
            raise NotImplementedError("only the norms RAFT-small uses at inference: instance | none")
        self.norm_fn = norm_fn
        self.conv1 = Conv2d(in_planes, planes // 4, 1)
        self.conv2 = Conv2d(planes // 4, planes // 4, 3, padding=1, stride=stride)
        self.conv3 = Conv2d(planes // 4, planes, 1)
        self.downsample = None
        if stride != 1:
            # key `downsample.0.*` as in nn.Sequential(conv, norm4); the norm has no parameters
            self.downsample = nn.Sequential(Conv2d(in_planes, planes, 1, stride=stride))

    def forward(self, x):
        if self.norm_fn == "instance":
            y = _inorm(self.conv1(x), 1)
            y = _inorm(self.conv2(y), 1)
            if self.downsample is not None:
                x = _inorm(self.downsample[0](x), 0)
            return _inorm(self.conv3(y), 2, res=x)          # relu(x + relu(norm(conv3)))
        y = self.conv1(x, act=ops.ACT_RELU)
        y = self.conv2(y, act=ops.ACT_RELU)
        if self.downsample is not None:
            x = self.downsample[0](x)
        return self.conv3(y, act=ops.ACT_RELU, res=x, res_mode=3)


class SmallEncoder(nn.Module):
    def __init__(self, output_dim=128, norm_fn="batch", dropout=0.0):
        super().__init__()
        self.norm_fn = norm_fn
        self.conv1 = Conv2d(3, 32, 7, stride=2, padding=3)
        self.in_planes = 32
        self.layer1 = self._make_layer(32, stride=1)
        self.layer2 = self._make_layer(64, stride=2)
        self.layer3 = self._make_layer(96, stride=2)
        self.conv2 = Conv2d(96, output_dim, 1)

    def _make_layer(self, dim, stride=1):
        layers = (BottleneckBlock(self.in_planes, dim, self.norm_fn, stride=stride), BottleneckBlock(dim, dim, self.norm_fn, stride=1))
        self.in_planes = dim
        return nn.Sequential(*layers)

    def forward(self, x, act=ops.ACT_NONE, act2=ops.ACT_NONE, act_split=0):
        is_list = isinstance(x, (tuple, list))
        if is_list:
            batch_dim = x[0].shape[0]
            x = torch.cat(x, dim=0)
        if self.norm_fn == "instance":
            x = _inorm(self.conv1(x), 1)
        else:
            x = self.conv1(x, act=ops.ACT_RELU)
        for layer in (self.layer1, self.layer2, self.layer3):
            for blk in layer:
                x = blk(x)
        x = self.conv2(x, act=act, act2=act2, act_split=act_split)
        if is_list:
            x = torch.split(x, [batch_dim, batch_dim], dim=0)
        return x
