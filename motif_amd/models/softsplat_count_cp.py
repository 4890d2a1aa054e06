"""Hit-count splatting (`/root/reference/models/softsplat_count_cp.py:14-52,163-175`): the input's
values are ignored (`new_ones`, :164); every in-bounds corner of every source pixel adds 1."""
import torch.nn as nn

from .. import ops


def FunctionSoftsplat(tenInput, tenFlow):
    return ops.splat(None, tenFlow, None, want=("cnt",))["cnt"]


class Softsplat_Count(nn.Module):
    def forward(self, img, flow):
        return FunctionSoftsplat(img, flow)
