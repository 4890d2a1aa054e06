"""Softmax splatting module on the fused HIP splat kernel.

Mirrors `/root/reference/models/softsplat_cp.py:320-357`: `Softsplat()(img, flow, z)` returns the
UN-normalised pair (sum of img*e^z*w, sum of e^z*w) -- the division is commented out in the reference
(:340-344) and happens after the two directions are added (`Ours.py:811-814`).
"""
import torch.nn as nn

from .. import ops


def FunctionSoftsplat(tenInput, tenFlow, tenMetric, strType):
    if strType != "softmax":
        raise NotImplementedError("MoTIF constructs Softsplat() with the default 'softmax' type")
    assert tenMetric is None or tenMetric.shape[1] == 1
    o = ops.splat(tenInput, tenFlow, tenMetric, want=("sum", "norm"))
    return o["sum"], o["norm"]


class Softsplat(nn.Module):
    def __init__(self, strType="softmax"):
        super().__init__()
        self.strType = strType

    def forward(self, img, flow, z):
        return FunctionSoftsplat(img, flow, z, self.strType)
