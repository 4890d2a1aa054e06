"""Parameter containers whose forward is a HIP kernel (state-dict keys match torch.nn's)."""
import math

import torch
import torch.nn as nn

from ... import ops


class Conv2d(nn.Module):
    """Same parameters/keys as nn.Conv2d (`weight`, `bias`); forward = fp32-MFMA implicit GEMM with a
    fused epilogue.  `x2` is concatenated to `x` on the channel axis inside the kernel."""

    def __init__(self, cin, cout, k, stride=1, padding=0, dilation=1, groups=1, bias=True, padding_mode="zeros"):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin // groups, k, k))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        self.stride, self.padding, self.dilation, self.groups = stride, padding, dilation, groups
        self.pad_mode = 1 if padding_mode == "reflect" else 0
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            nn.init.zeros_(self.bias)
        self._plan = None

    def plan(self):
        if self._plan is None or self._plan.weight is not self.weight:
            self._plan = ops.ConvPlan(self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups, self.pad_mode)
        return self._plan

    def forward(self, x, x2=None, act=ops.ACT_NONE, res=None, res_mode=0, act2=ops.ACT_NONE, act_split=0, out=None):
        return ops.conv2d(self.plan(), x, x2, act, res, res_mode, act2, act_split, out)


class Linear(nn.Module):
    """nn.Linear-shaped parameter holder for the SIREN stacks (consumed packed by the MLP kernels)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        nn.init.zeros_(self.bias)
