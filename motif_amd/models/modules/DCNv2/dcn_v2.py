"""Modulated deformable conv v2 modules on the HIP kernel.

Mirrors `/root/reference/models/modules/DCNv2/dcn_v2.py:15-47` (`dcn_v2_conv`), `:53-84` (DCNv2) and
`:110-140` (DCN_sep): same constructors, parameters and state-dict keys (`weight`, `bias`,
`conv_offset_mask.{weight,bias}`).  Forward only (inference path); no CPU implementation, like the
reference (`src/dcn_v2.h:38`).
"""
import math

import torch
import torch.nn as nn

from .... import ops
from ..layers import Conv2d


def dcn_v2_conv(input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups):
    """Operator form of `_DCNv2.apply` -> `_ext.dcn_v2_forward` (dcn_v2.py:24-27)."""
    st = stride[0] if isinstance(stride, (tuple, list)) else stride
    pd = padding[0] if isinstance(padding, (tuple, list)) else padding
    dl = dilation[0] if isinstance(dilation, (tuple, list)) else dilation
    kh, kw = weight.shape[2:]
    return ops.dcn_v2_raw(input, offset, mask, weight, bias, kh, kw, st, pd, dl, deformable_groups)


class DCNv2(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = (k, k)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, k, k))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        stdv = 1.0 / math.sqrt(in_channels * k * k)
        self.weight.data.uniform_(-stdv, stdv)
        self._dplan = None

    def dplan(self):
        if self._dplan is None or self._dplan.weight is not self.weight:
            self._dplan = ops.DcnPlan(self.weight, self.bias)
        return self._dplan

    def forward(self, input, offset, mask):
        return dcn_v2_conv(input, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation, self.deformable_groups)


class DCN_sep(DCNv2):
    """Offsets and masks are generated from other features (`fea`)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, deformable_groups)
        k = self.kernel_size[0]
        self.conv_offset_mask = Conv2d(in_channels, deformable_groups * 3 * k * k, k, stride=stride, padding=padding)
        nn.init.zeros_(self.conv_offset_mask.weight)

    def forward(self, input, fea, act=ops.ACT_NONE):
        k = self.kernel_size[0]
        dg = self.deformable_groups
        # chunk(3)+cat(o1,o2) keeps the channel order: first 2*dg*k*k channels are offsets, the rest
        # the mask logits -> sigmoid fused into the conv epilogue for that channel range
        om = self.conv_offset_mask(fea, act=ops.ACT_NONE, act2=ops.ACT_SIGMOID, act_split=2 * dg * k * k)
        return ops.dcn_v2(self.dplan(), input, om, dg, act, k, k, self.stride, self.padding, self.dilation)
