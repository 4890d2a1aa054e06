"""Max splatting (`/root/reference/models/softsplat_max_cp.py:12-58,254,337-352`): atomic max of
img*w over the four corners, output initialised to ones."""
import torch.nn as nn

from .. import ops


def FunctionSoftsplat(tenInput, tenFlow):
    if tenInput.shape[1] != 1:
        raise NotImplementedError("MoTIF max-splats the single-channel e^z map (Ours.py:805)")
    return ops.splat(tenInput, tenFlow, None, want=("max",))["max"]


class Softsplat_Max(nn.Module):
    def forward(self, img, flow):
        return FunctionSoftsplat(img, flow)
